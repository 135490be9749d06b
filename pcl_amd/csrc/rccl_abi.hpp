// rccl_abi.hpp -- the few constants of RCCL's C ABI that icp_loop.hip passes by value.  libpclhip.so binds RCCL at run time
// (weak references / dlopen: single-GPU users need no RCCL, and a process that already carries one keeps it), so it cannot
// include <rccl/rccl.h> next to its own opaque declarations of the entry points.  rccl_abi_check.cpp DOES include the
// header of the ROCm it is built with and static_asserts every value and signature below against it at build time; at run
// time the library asks ncclGetVersion and refuses anything but major version 2 (the ABI these were taken from).
// Taken from rccl.h of RCCL 2.27.7 (ROCm 7.2; unchanged since NCCL 2.10 introduced ncclBfloat16 behind them).
#pragma once
#include <cstddef>

namespace pclhip {
namespace rccl_abi {
constexpr int kUniqueIdBytes = 128;  // NCCL_UNIQUE_ID_BYTES, sizeof(ncclUniqueId)
constexpr int kUint64 = 5;           // ncclDataType_t::ncclUint64
constexpr int kFloat64 = 8;          // ncclDataType_t::ncclFloat64 (= ncclDouble)
constexpr int kSum = 0;              // ncclRedOp_t::ncclSum
constexpr int kMin = 3;              // ncclRedOp_t::ncclMin
constexpr int kSuccess = 0;          // ncclResult_t::ncclSuccess
constexpr int kMajor = 2;            // NCCL_MAJOR the constants belong to
}  // namespace rccl_abi
}  // namespace pclhip
