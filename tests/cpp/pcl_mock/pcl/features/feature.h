// forwarding header of the PCL mock (tests/cpp/pcl_mock/pcl_mock.hpp), at PCL's include path
#pragma once
#include "../../pcl_mock.hpp"
