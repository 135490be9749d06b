// rccl_abi_check.cpp -- build-time check of rccl_abi.hpp against the RCCL header of this ROCm.  No code: a mismatch stops
// the build of libpclhip.so, instead of the first 8-GPU run finding it as undefined behaviour.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <type_traits>

#include "rccl_abi.hpp"

namespace abi = pclhip::rccl_abi;
static_assert(NCCL_MAJOR == abi::kMajor, "rccl_abi.hpp was written for another major version of RCCL");
static_assert(sizeof(ncclUniqueId) == abi::kUniqueIdBytes && NCCL_UNIQUE_ID_BYTES == abi::kUniqueIdBytes, "ncclUniqueId");
static_assert(int(ncclUint64) == abi::kUint64, "ncclUint64");
static_assert(int(ncclFloat64) == abi::kFloat64 && int(ncclDouble) == abi::kFloat64, "ncclFloat64");
static_assert(int(ncclSum) == abi::kSum, "ncclSum");
static_assert(int(ncclMin) == abi::kMin, "ncclMin");
static_assert(int(ncclSuccess) == abi::kSuccess, "ncclSuccess");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int),
              "RCCL's enums travel as int");
static_assert(std::is_pointer<ncclComm_t>::value, "ncclComm_t is an opaque pointer");
// the signatures icp_loop.hip casts its weak references / dlsym results to
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t,
                                                                     ncclComm_t, hipStream_t)>::value, "ncclAllReduce");
static_assert(std::is_same<decltype(&ncclGetVersion), ncclResult_t (*)(int*)>::value, "ncclGetVersion");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString");
