// icp_loop.hip -- the device-driven ICP loop and the multi-GPU record exchange.
//
// IterativeClosestPoint::computeTransformation (registration/include/pcl/registration/impl/icp.hpp:113-268)
// alternates a data-parallel part (transform, correspondences, normal-system sums) with a tiny serial part
// (6x6 solve, final = T * final, DefaultConvergenceCriteria).  Doing the serial part on the host costs a
// read-back, a host solve and a launch per iteration (~60 us, 8 % of a converged 10M-point iteration in
// round 1).  Here the serial part is icp_solve_kernel (search.hip) and the loop state lives in device memory
// (IcpControl): the kernels of consecutive iterations are queued back to back on the context's stream, the
// one after a finished alignment falls through (`stop`), and the host only reads the step records the solve
// kernel leaves in pinned memory -- one iteration behind the GPU, so it never stalls it.
//
// Multi-GPU: the 32-double record is summed over the ranks between the reduction and the solve, on the same
// stream -- ncclAllReduce through a communicator created here (RCCL over xGMI), or the caller's hook.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pclhip_internal.hpp"
#include "rccl_abi.hpp"  // the by-value constants of RCCL's ABI; checked against <rccl/rccl.h> by rccl_abi_check.cpp

using namespace pclhip;

// ---------------------------------------------------------------------------------------------------
// RCCL, bound at run time: libpclhip.so has no link-time dependency on it (single-GPU users need none), and a
// process that already carries an RCCL (torch ships its own) keeps using that one instead of a second copy.
// ---------------------------------------------------------------------------------------------------
// RCCL's entry points as WEAK references: null unless the link unit this library ends up in (or a library already loaded
// into the global scope) defines them -- then they are used directly; otherwise librccl is dlopen()ed below.  (The CPU
// emulation of the test tier links a stand-in under these names: tests/wavesim/wavesim_rt.cpp -- a link-time hook, no
// conditional compilation here.)  Declared with opaque signatures: only their addresses are taken.
extern "C" {
__attribute__((weak)) void ncclGetUniqueId();
__attribute__((weak)) void ncclCommInitRank();
__attribute__((weak)) void ncclCommDestroy();
__attribute__((weak)) void ncclAllReduce();
__attribute__((weak)) void ncclGetVersion();
}
static void (*const pclhip_weak_ncclGetUniqueId)() = &ncclGetUniqueId;
static void (*const pclhip_weak_ncclCommInitRank)() = &ncclCommInitRank;
static void (*const pclhip_weak_ncclCommDestroy)() = &ncclCommDestroy;
static void (*const pclhip_weak_ncclAllReduce)() = &ncclAllReduce;
static void (*const pclhip_weak_ncclGetVersion)() = &ncclGetVersion;

namespace {

struct NcclUniqueId {
  char internal[rccl_abi::kUniqueIdBytes];  // ncclUniqueId
};
typedef void* NcclComm;

struct RcclApi {
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  bool ok = false;
  std::string why;
};

// The constants of rccl_abi.hpp belong to NCCL/RCCL major version 2: whatever library the entry points resolved to (this
// process may carry another build than the ROCm this was compiled with) must say it is one.  A library without
// ncclGetVersion is accepted only when it was LINKED in (the stand-in of the test tier's emulation).
void check_rccl_version(RcclApi& a, bool linked_in) {
  if (!a.ok) return;
  if (a.GetVersion == nullptr) {
    if (!linked_in) {
      a.ok = false;
      a.why = "the RCCL library has no ncclGetVersion: its ABI cannot be checked";
    }
    return;
  }
  int v = 0;
  if (a.GetVersion(&v) != rccl_abi::kSuccess) {
    a.ok = false;
    a.why = "ncclGetVersion failed";
    return;
  }
  const int major = v >= 10000 ? v / 10000 : v / 1000;  // NCCL_VERSION: major * 10000 + ... since 2.9, major * 1000 + ... before
  if (major != rccl_abi::kMajor) {
    a.ok = false;
    a.why = "RCCL version " + std::to_string(v) + " is not major version " + std::to_string(rccl_abi::kMajor) +
            ": the enum values this library passes were not checked against it";
  }
}

RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    // linked in already (whatever this library is part of provides the entry points): use them as they are
    if (pclhip_weak_ncclGetUniqueId != nullptr && pclhip_weak_ncclCommInitRank != nullptr &&
        pclhip_weak_ncclCommDestroy != nullptr && pclhip_weak_ncclAllReduce != nullptr) {
      a.GetUniqueId = reinterpret_cast<int (*)(NcclUniqueId*)>(pclhip_weak_ncclGetUniqueId);
      a.CommInitRank = reinterpret_cast<int (*)(NcclComm*, int, NcclUniqueId, int)>(pclhip_weak_ncclCommInitRank);
      a.CommDestroy = reinterpret_cast<int (*)(NcclComm)>(pclhip_weak_ncclCommDestroy);
      a.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t)>(pclhip_weak_ncclAllReduce);
      if (pclhip_weak_ncclGetVersion != nullptr) a.GetVersion = reinterpret_cast<int (*)(int*)>(pclhip_weak_ncclGetVersion);
      a.ok = true;
      check_rccl_version(a, true);
      return a;
    }
    void* h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclCommInitRank") == nullptr) {
      const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
      for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
      if (!h) {
        a.why = "RCCL not found (librccl.so): multi-GPU registration needs it";
        return a;
      }
    }
    auto sym = [&](const char* n) -> void* {
      void* p = h ? dlsym(h, n) : nullptr;
      return p ? p : dlsym(RTLD_DEFAULT, n);
    };
    a.GetUniqueId = reinterpret_cast<int (*)(NcclUniqueId*)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<int (*)(NcclComm*, int, NcclUniqueId, int)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<int (*)(NcclComm)>(sym("ncclCommDestroy"));
    a.AllReduce =
        reinterpret_cast<int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
    a.GetVersion = reinterpret_cast<int (*)(int*)>(sym("ncclGetVersion"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    if (!a.ok) a.why = "the RCCL library lacks a required symbol";
    check_rccl_version(a, false);
    return a;
  }();
  return api;
}

std::string nccl_error(int rc) {
  const RcclApi& a = rccl();
  return a.GetErrorString ? std::string(a.GetErrorString(rc)) : ("nccl error " + std::to_string(rc));
}

}  // namespace

struct pclhip_comm {
  pclhip_ctx* ctx = nullptr;
  NcclComm comm = nullptr;
  int rank = 0, nranks = 1;
};

extern "C" {

pclhip_status pclhip_comm_get_unique_id(unsigned char id[PCLHIP_COMM_ID_BYTES]) {
  if (!id) return PCLHIP_ERR_INVALID;
  RcclApi& a = rccl();
  if (!a.ok) {
    set_error(nullptr, a.why);
    return PCLHIP_ERR_STATE;
  }
  NcclUniqueId u;
  const int rc = a.GetUniqueId(&u);
  if (rc != 0) {
    set_error(nullptr, "ncclGetUniqueId: " + nccl_error(rc));
    return PCLHIP_ERR_HIP;
  }
  static_assert(sizeof(NcclUniqueId) == PCLHIP_COMM_ID_BYTES && rccl_abi::kUniqueIdBytes == PCLHIP_COMM_ID_BYTES, "id size");
  std::memcpy(id, &u, sizeof u);
  return PCLHIP_OK;
}

pclhip_status pclhip_comm_create(pclhip_ctx* ctx, int rank, int nranks, const unsigned char id[PCLHIP_COMM_ID_BYTES],
                                 pclhip_comm** out) {
  if (!ctx || !out || !id) return PCLHIP_ERR_INVALID;
  *out = nullptr;
  PCLHIP_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks, "rank out of range");
  RcclApi& a = rccl();
  if (!a.ok) {
    set_error(ctx, a.why);
    return PCLHIP_ERR_STATE;
  }
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  NcclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  NcclComm c = nullptr;
  const int rc = a.CommInitRank(&c, nranks, u, rank);
  if (rc != 0) {
    set_error(ctx, "ncclCommInitRank: " + nccl_error(rc));
    return PCLHIP_ERR_HIP;
  }
  pclhip_comm* cm = new pclhip_comm();
  cm->ctx = ctx;
  cm->comm = c;
  cm->rank = rank;
  cm->nranks = nranks;
  *out = cm;
  return PCLHIP_OK;
}

void pclhip_comm_destroy(pclhip_comm* comm) {
  if (!comm) return;
  if (comm->ctx) {
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
  }
  if (comm->comm) (void)rccl().CommDestroy(comm->comm);
  delete comm;
}

int pclhip_comm_rank(const pclhip_comm* comm) { return comm ? comm->rank : 0; }
int pclhip_comm_size(const pclhip_comm* comm) { return comm ? comm->nranks : 1; }

pclhip_status pclhip_comm_allreduce_sum_f64(pclhip_comm* comm, double* device_buf, int count) {
  if (!comm || !device_buf || count < 0) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = comm->ctx;
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  const int rc = rccl().AllReduce(device_buf, device_buf, size_t(count), rccl_abi::kFloat64, rccl_abi::kSum, comm->comm, ctx->stream);
  if (rc != 0) {
    set_error(ctx, "ncclAllReduce: " + nccl_error(rc));
    return PCLHIP_ERR_HIP;
  }
  return PCLHIP_OK;
}

pclhip_status pclhip_icp_set_comm(pclhip_icp* icp, pclhip_comm* comm) {
  if (!icp) return PCLHIP_ERR_INVALID;
  PCLHIP_REQUIRE(icp->ctx, comm == nullptr || comm->ctx == icp->ctx, "communicator and registration must share a context");
  icp->comm = comm;
  return PCLHIP_OK;
}

}  // extern "C"

namespace pclhip {

// sum of `count` doubles in device memory over the ranks, on the context's stream (the native communicator or the
// caller's hook; nothing to do for a registration on one GPU)
pclhip_status allreduce_doubles(pclhip_icp* icp, double* device_buf, int count) {
  pclhip_ctx* ctx = icp->ctx;
  if (icp->comm != nullptr) return pclhip_comm_allreduce_sum_f64(icp->comm, device_buf, count);
  if (icp->allreduce != nullptr) {
    const int rc = icp->allreduce(icp->allreduce_user, device_buf, count, ctx->stream);
    if (rc != 0) {
      set_error(ctx, "all-reduce hook failed");
      return PCLHIP_ERR_STATE;
    }
  }
  return PCLHIP_OK;
}

pclhip_status allreduce_record(pclhip_icp* icp) { return allreduce_doubles(icp, icp->sums_dev, PCLHIP_ICP_NSUMS); }

// element-wise MINIMUM of 64-bit keys over the ranks (the OneToOne rejector under target sharding: per target point the
// smallest (distance, query) key any rank holds).  Native communicator only: the caller's hook sums doubles.
pclhip_status allreduce_min_u64(pclhip_icp* icp, unsigned long long* device_buf, size_t count) {
  pclhip_ctx* ctx = icp->ctx;
  if (icp->comm == nullptr) {
    set_error(ctx, "a minimum over the ranks needs the native communicator (pclhip_icp_set_comm)");
    return PCLHIP_ERR_STATE;
  }
  const int rc = rccl().AllReduce(device_buf, device_buf, count, rccl_abi::kUint64, rccl_abi::kMin, icp->comm->comm, ctx->stream);
  if (rc != 0) {
    set_error(ctx, "ncclAllReduce(min): " + nccl_error(rc));
    return PCLHIP_ERR_HIP;
  }
  return PCLHIP_OK;
}

// OneToOne under target sharding indexes its per-target keys by the ORIGINAL index of the whole target cloud: every rank
// must have indexed that same cloud (through its own subset list).  min and max of n_orig over the ranks in one collective
// (min of {n, ~n}); a mismatch is an error on every rank, before the big all-reduce could hang or corrupt.  Once per
// (registration, communicator).
pclhip_status check_same_target_size(pclhip_icp* icp) {
  pclhip_ctx* ctx = icp->ctx;
  if (icp->comm == nullptr || icp->target_size_checked_for == icp->comm) return PCLHIP_OK;
  unsigned long long* d = nullptr;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &d, 16));
  const unsigned long long n = icp->target->n_orig;
  unsigned long long h[2] = {n, ~n};
  pclhip_status st = PCLHIP_OK;
  hipError_t e = hipMemcpyAsync(d, h, 16, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // h lives on this stack
  if (e == hipSuccess) st = allreduce_min_u64(icp, d, 2);
  if (e == hipSuccess && st == PCLHIP_OK) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && st == PCLHIP_OK) e = hipStreamSynchronize(ctx->stream);
  (void)dev_free(ctx, d);
  if (st != PCLHIP_OK) return st;
  PCLHIP_CHECK_HIP(ctx, e);
  if (h[0] != n || ~h[1] != n) {
    set_error(ctx, "OneToOne under target sharding: the ranks indexed target clouds of different sizes (" +
                       std::to_string(h[0]) + " .. " + std::to_string(~h[1]) +
                       " points): every rank must build its index over the WHOLE cloud plus its subset list");
    return PCLHIP_ERR_STATE;
  }
  icp->target_size_checked_for = icp->comm;
  return PCLHIP_OK;
}

bool icp_is_sharded(const pclhip_icp* icp) { return icp->comm != nullptr || icp->allreduce != nullptr; }

// ---------------------------------------------------------------------------------------------------
// loop state
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kRing = 64;  // step records / event quadruples in flight at most

pclhip_status ensure_loop_state(pclhip_icp* icp) {
  pclhip_ctx* ctx = icp->ctx;
  if (icp->ctl != nullptr) return PCLHIP_OK;
  PCLHIP_CHECK_HIP(ctx, dev_malloc(ctx, &icp->ctl, sizeof(IcpControl)));
  PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &icp->ctl_host, sizeof(IcpControl)));
  PCLHIP_CHECK_HIP(ctx, pinned_malloc(ctx, &icp->steps, sizeof(IcpStepRecord) * kRing));
  std::memset(icp->steps, 0, sizeof(IcpStepRecord) * kRing);
  icp->steps_capacity = kRing;
  icp->step_events.assign(size_t(kRing) * 4, nullptr);
  for (int i = 0; i < kRing; ++i) {
    // markers between kernels of one stream: device-scope release is enough; the last one of a step
    // publishes the step record to the host, so it releases to the system
    for (int e = 0; e < 4; ++e)
      PCLHIP_CHECK_HIP(ctx, hipEventCreateWithFlags(&icp->step_events[size_t(i) * 4 + e],
                                                    e == 3 ? hipEventDefault : hipEventReleaseToDevice));
  }
  return PCLHIP_OK;
}

void fill_criteria(const pclhip_icp_params* p, cf::Criteria& c) {  // impl/icp.hpp:157-161
  c.max_iterations = p->max_iterations;
  c.failure_after_max_iterations = p->failure_after_max_iterations;
  c.max_iterations_similar_transforms = p->max_iterations_similar_transforms;
  c.min_number_correspondences = p->min_number_correspondences;
  c.rotation_threshold = p->transformation_rotation_epsilon > 0 ? p->transformation_rotation_epsilon : 0.99999;
  c.translation_threshold = p->transformation_epsilon;
  c.mse_threshold_relative = p->euclidean_fitness_epsilon;
  c.mse_threshold_absolute = p->mse_threshold_absolute;
}

const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

// upload the control block: an alignment starts with the next launch
pclhip_status arm_loop(pclhip_icp* icp, const pclhip_icp_params* p, const float* guess, bool auto_restart) {
  pclhip_ctx* ctx = icp->ctx;
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // nothing of a previous loop is still reading the staging copy
  IcpControl& c = *icp->ctl_host;
  std::memset(&c, 0, sizeof c);
  const float* g = guess ? guess : kIdentity;
  std::memcpy(c.guess, g, sizeof c.guess);
  std::memcpy(c.final_T, g, sizeof c.final_T);  // icp.hpp:123
  std::memcpy(c.T_apply, g, sizeof c.T_apply);  // :126-131, applied by the first launch
  std::memcpy(c.Tk, kIdentity, sizeof c.Tk);
  c.restart = 1;
  c.stop = 0;
  c.mode = p->mode;
  c.auto_restart = auto_restart ? 1 : 0;
  c.nr_iterations = 0;
  c.step = 0;
  c.log_capacity = icp->steps_capacity;
  fill_criteria(p, c.crit);
  if (auto_restart) {
    // A stream of steps behaves like a fresh registration object aligned again and again: the memory starts
    // empty and persists from one alignment of the stream to the next, but not across calls -- a call boundary
    // may cut an alignment short, and resuming with "previous MSE = this very iteration's MSE" would end every
    // following alignment at its first iteration (ABS_MSE).
    c.st.prev_mse = DBL_MAX;
    c.st.iterations_similar_transforms = 0;
    c.st.convergence_state = cf::NOT_CONVERGED;
  } else {
    c.st.prev_mse = icp->prev_mse;  // the criteria's memory persists across align() calls, as in the reference
    c.st.iterations_similar_transforms = icp->iterations_similar_transforms;
    c.st.convergence_state = icp->convergence_state;
  }
  PCLHIP_CHECK_HIP(ctx, hipMemcpyAsync(icp->ctl, icp->ctl_host, sizeof c, hipMemcpyHostToDevice, ctx->stream));
  return PCLHIP_OK;
}

struct StepTimes {
  float search_ms = 0, kernels_ms = 0, step_ms = 0;
};

// Runs the queue: at most `window` iterations are in flight; `on_step` sees every completed iteration in
// order and returns false to stop feeding the queue (the alignment ended / enough steps).
template <class F>
pclhip_status run_loop(pclhip_icp* icp, const pclhip_icp_params* p, int window, long max_steps, F on_step) {
  pclhip_ctx* ctx = icp->ctx;
  const double md2 = p->max_correspondence_distance * p->max_correspondence_distance;
  const bool use_max = md2 < double(FLT_MAX);
  float fmax2 = FLT_MAX;
  if (use_max) {  // largest float <= md2 (correspondence_estimation.hpp:161,176 compares in double)
    fmax2 = float(md2);
    if (double(fmax2) > md2) fmax2 = std::nextafterf(fmax2, 0.0f);
  }
  if (window < 1) window = 1;
  if (window > icp->steps_capacity - 1) window = icp->steps_capacity - 1;
  long enq = 0, done = 0;
  bool feed = true;
  pclhip_status st = PCLHIP_OK;
  while (done < enq || (feed && (max_steps < 0 || enq < max_steps))) {
    while (feed && (max_steps < 0 || enq < max_steps) && enq - done < window) {
      hipEvent_t* ev = &icp->step_events[size_t(enq % icp->steps_capacity) * 4];
      st = launch_icp_iterate(icp, nullptr, fmax2, use_max, p->mode, ev);
      if (st != PCLHIP_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return st;
      }
      ++enq;
    }
    if (done == enq) break;
    hipEvent_t* ev = &icp->step_events[size_t(done % icp->steps_capacity) * 4];
    PCLHIP_CHECK_HIP(ctx, hipEventSynchronize(ev[3]));
    const IcpStepRecord rec = icp->steps[done % icp->steps_capacity];
    StepTimes t;
    if (feed) {  // launches queued behind a finished alignment fell through: no record, no times
      if (rec.step != int(done)) {
        set_error(ctx, "ICP loop: step record out of sequence");
        (void)hipStreamSynchronize(ctx->stream);
        return PCLHIP_ERR_STATE;
      }
      (void)hipEventElapsedTime(&t.search_ms, ev[0], ev[1]);
      (void)hipEventElapsedTime(&t.kernels_ms, ev[0], ev[2]);
      (void)hipEventElapsedTime(&t.step_ms, ev[0], ev[3]);
      if (!on_step(rec, t)) feed = false;
    }
    ++done;
  }
  return PCLHIP_OK;
}

// iterations in flight: the one the host waits for + those queued ahead of its knowledge (option "icp_lookahead")
int loop_window(const pclhip_ctx* ctx) { return (ctx->opt_lookahead < 0 ? 0 : ctx->opt_lookahead) + 1; }

}  // namespace

// IterativeClosestPoint::computeTransformation without rejectors / reciprocal correspondences
pclhip_status icp_align_device(pclhip_icp* icp, const pclhip_icp_params* params, const float* guess,
                               pclhip_icp_result* res) {
  pclhip_ctx* ctx = icp->ctx;
  pclhip_status st = ensure_loop_state(icp);
  if (st != PCLHIP_OK) return st;
  st = arm_loop(icp, params, guess, false);
  if (st != PCLHIP_OK) return st;
  std::memcpy(res->final_transformation, guess ? guess : kIdentity, sizeof res->final_transformation);
  std::memcpy(res->last_transformation, kIdentity, sizeof res->last_transformation);
  double search_ms = 0, total_ms = 0;
  st = run_loop(icp, params, loop_window(icp->ctx), -1, [&](const IcpStepRecord& r, const StepTimes& t) {
    search_ms += t.kernels_ms;
    total_ms += t.step_ms;
    icp->last_kernel_ms = t.kernels_ms;
    icp->last_search_ms = t.search_ms;
    res->num_correspondences = uint64_t(r.num_correspondences);
    res->nr_iterations = r.iteration;
    res->convergence_state = r.convergence_state;
    res->converged = r.converged;
    icp->prev_mse = r.prev_mse;
    icp->iterations_similar_transforms = r.similar;
    icp->convergence_state = r.convergence_state;
    if (r.convergence_state != cf::NO_CORRESPONDENCES) {
      res->mse = r.mse;
      std::memcpy(res->final_transformation, r.final_T, sizeof r.final_T);
      std::memcpy(res->last_transformation, r.Tk, sizeof r.Tk);
    }
    return r.ended == 0;
  });
  if (st == PCLHIP_OK) st = owned_groups_catch_up(icp, params->mode);  // target sharding: see search.hip, served groups
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (st != PCLHIP_OK) return st;
  res->gpu_ms = total_ms;
  res->gpu_ms_search_kernel = search_ms;
  return PCLHIP_OK;
}

}  // namespace pclhip

extern "C" void pclhip_convergence_init(pclhip_convergence_state* st) {
  if (!st) return;
  st->prev_mse = DBL_MAX;  // default_convergence_criteria.h: correspondences_prev_mse_
  st->iterations_similar_transforms = 0;
  st->convergence_state = cf::NOT_CONVERGED;
}

extern "C" int pclhip_convergence_has_converged(const pclhip_icp_params* params, pclhip_convergence_state* st,
                                                int nr_iterations, const float T[16], double mse) {
  if (!params || !st || !T) return 0;
  cf::Criteria c;
  fill_criteria(params, c);
  cf::CriteriaState s;
  s.prev_mse = st->prev_mse;
  s.iterations_similar_transforms = st->iterations_similar_transforms;
  s.convergence_state = st->convergence_state;
  const bool r = cf::has_converged(c, s, nr_iterations, T, mse);
  st->prev_mse = s.prev_mse;
  st->iterations_similar_transforms = s.iterations_similar_transforms;
  st->convergence_state = s.convergence_state;
  return r ? 1 : 0;
}

extern "C" pclhip_status pclhip_icp_run_steps(pclhip_icp* icp, const pclhip_icp_params* params, const float* guess,
                                              int n_steps, pclhip_icp_step* out_steps) {
  if (!icp || !params || n_steps < 0 || (n_steps > 0 && !out_steps)) return PCLHIP_ERR_INVALID;
  pclhip_ctx* ctx = icp->ctx;
  PCLHIP_REQUIRE(ctx, icp->src_cur != nullptr, "no source cloud set");
  {
    const pclhip_status sf = sharded_filters_ok(icp);
    if (sf != PCLHIP_OK) return sf;
  }
  if (params->mode != PCLHIP_ICP_POINT_TO_POINT && !icp->target->has_normals) {
    set_error(ctx, "point-to-plane ICP needs target normals (pclhip_normals / pclhip_index_set_normals)");
    return PCLHIP_ERR_STATE;
  }
  if (params->mode == PCLHIP_ICP_SYMMETRIC && icp->src_nrm_cur == nullptr) {
    set_error(ctx, "the symmetric objective needs source normals (pclhip_icp_set_source_normals)");
    return PCLHIP_ERR_STATE;
  }
  PCLHIP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  pclhip_status st = ensure_loop_state(icp);
  if (st != PCLHIP_OK) return st;
  st = arm_loop(icp, params, guess, true);
  if (st != PCLHIP_OK) return st;
  int got = 0;
  // every step is queued before the first result is looked at (up to the ring size): the GPU runs the
  // iterations back to back, alignments restart on the device
  st = run_loop(icp, params, icp->steps_capacity - 1, n_steps, [&](const IcpStepRecord& r, const StepTimes& t) {
    pclhip_icp_step& o = out_steps[got++];
    o.iteration = r.iteration;
    o.convergence_state = r.convergence_state;
    o.converged = r.converged;
    o.alignment_ended = r.ended;
    o.num_correspondences = uint64_t(r.num_correspondences);
    o.mse = r.mse;
    o.search_ms = t.search_ms;
    o.kernels_ms = t.kernels_ms;
    o.step_ms = t.step_ms;
    std::memcpy(o.final_transformation, r.final_T, sizeof r.final_T);
    icp->last_kernel_ms = t.kernels_ms;  // (the criteria memory of align() is left alone, see arm_loop)
    icp->last_search_ms = t.search_ms;
    return true;
  });
  if (st == PCLHIP_OK) st = owned_groups_catch_up(icp, params->mode);
  PCLHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return st;
}
