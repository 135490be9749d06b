"""pcl_amd -- MI355X-native ICP hot path (k-NN correspondences, normals, point-to-plane ICP,
VoxelGrid) behind PCL's plugin surface.  Compute lives in libpclhip.so (pcl_amd/csrc, HIP/gfx950)
behind the C ABI of include/pclhip.h; this package is the ctypes host mirror used by tests and
bench.py.  The C++ mirror of the same surface is include/pclhip/pcl_compat.hpp."""
from . import synth  # noqa: F401
from ._lib import (POINT_TO_PLANE, POINT_TO_POINT, SYMMETRIC, PclHipError, PclHipUnavailable)  # noqa: F401
from .api import (Communicator, Context, CorrespondenceEstimation, CorrespondenceRejectorDistance,  # noqa: F401
                  CorrespondenceRejectorMedianDistance, CorrespondenceRejectorOneToOne,
                  CorrespondenceRejectorTrimmed, DefaultConvergenceCriteria, IterativeClosestPoint,
                  IterativeClosestPointWithNormals, KdTree, NormalEstimation, VoxelGrid,
                  default_context, estimateRigidTransformation, getPCDHeader, loadPCDField, loadPCDFile, savePCDFile)
