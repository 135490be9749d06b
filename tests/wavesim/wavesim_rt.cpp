// wavesim_rt.cpp -- the runtime half of tests/wavesim (TEST INFRASTRUCTURE, see wavesim.hpp): fibers and the
// rendezvous of cross-lane operations, a pool of OS threads that runs the workgroups of a launch, and the few dozen HIP
// runtime entry points the host side of libpclhip calls, implemented over plain host memory ("device" allocations are
// malloc'ed blocks remembered in a table so that hipPointerGetAttributes can tell them from user memory; streams are
// immediate; events are timestamps).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace wavesim {

struct Idx3 {
  unsigned x, y, z;
};
struct LaneCtx {
  Idx3 tid, bid, bdim, gdim;
};
thread_local LaneCtx* cur = nullptr;

namespace {

extern "C" void wavesim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl wavesim_switch
.type wavesim_switch,@function
wavesim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size wavesim_switch,.-wavesim_switch
)");

enum State { READY, WAIT_WAVE, WAIT_BLOCK, DONE };
constexpr size_t STACK_BYTES = 256 << 10;
constexpr unsigned MAX_LANES = 1024;

struct Wave {
  uint64_t val[64];  // operands of the lanes of the last release (read by each of them right after it resumes)
};
struct Lane {
  LaneCtx ctx;
  void* sp = nullptr;
  State state = DONE;
  const char* site = nullptr;
  uint64_t pending = 0;  // operand of the cross-lane operation the lane waits at
  uint64_t live = 0;     // lanes that took part in the lane's last cross-lane operation
  int depth = 0;         // nesting of lane-masked regions (PCLHIP_LANE_MASKED_REGION) the lane is inside of
  Wave* wave = nullptr;
};

struct Worker {  // per OS thread
  char* stacks = nullptr;
  std::vector<Lane> lanes;
  std::vector<Wave> waves;
  void* sched_sp = nullptr;
  Lane* running = nullptr;
  const std::function<void()>* body = nullptr;
};
thread_local Worker* tw = nullptr;

[[noreturn]] void die(const char* what, const char* a = "", const char* b = "") {
  std::fprintf(stderr, "wavesim: %s %s %s\n", what, a ? a : "(null)", b ? b : "(null)");
  std::fflush(stderr);
  std::abort();
}

void yield_to_scheduler() {
  Worker* w = tw;
  Lane* me = w->running;
  wavesim_switch(&me->sp, w->sched_sp);
}

extern "C" void wavesim_fiber_entry() {
  Worker* w = tw;
  Lane* me = w->running;
  (*w->body)();
  me->state = DONE;
  yield_to_scheduler();
  die("a finished fiber was resumed");
}

void resume(Worker* w, Lane* l) {
  w->running = l;
  cur = &l->ctx;
  wavesim_switch(&w->sched_sp, l->sp);
  w->running = nullptr;
}

void run_block(Worker* w, dim3 grid, dim3 block, unsigned bx, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads > MAX_LANES) die("block too large");
  if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) die("only 1-D launches are emulated");
  if (!w->stacks) {
    w->stacks = static_cast<char*>(mmap(nullptr, size_t(MAX_LANES) * STACK_BYTES, PROT_READ | PROT_WRITE,
                                        MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (w->stacks == MAP_FAILED) die("cannot map fiber stacks");
    w->lanes.resize(MAX_LANES);
    w->waves.resize(MAX_LANES / 64);
  }
  w->body = &body;
  const unsigned nwaves = (nthreads + 63) / 64;
  for (unsigned t = 0; t < nthreads; ++t) {
    Lane& l = w->lanes[t];
    l.ctx.tid = {t, 0, 0};
    l.ctx.bid = {bx, 0, 0};
    l.ctx.bdim = {block.x, 1, 1};
    l.ctx.gdim = {grid.x, 1, 1};
    l.state = READY;
    l.site = nullptr;
    l.depth = 0;
    l.wave = &w->waves[t / 64];
    // initial frame: six callee-saved registers, the entry point, a null return address (see wavesim_switch)
    uintptr_t top = reinterpret_cast<uintptr_t>(w->stacks + size_t(t + 1) * STACK_BYTES) & ~uintptr_t(15);
    void** f = reinterpret_cast<void**>(top - 64);
    for (int i = 0; i < 6; ++i) f[i] = nullptr;
    f[6] = reinterpret_cast<void*>(&wavesim_fiber_entry);
    f[7] = nullptr;
    l.sp = f;
  }
  unsigned done = 0;
  while (done < nthreads) {
    bool progress = false;
    for (unsigned wv = 0; wv < nwaves; ++wv) {
      const unsigned t0 = wv * 64, t1 = std::min(nthreads, t0 + 64);
      for (;;) {
        bool ran = false;
        for (unsigned t = t0; t < t1; ++t) {
          Lane& l = w->lanes[t];
          if (l.state != READY) continue;
          resume(w, &l);
          ran = true;
          if (l.state == DONE) ++done;
        }
        if (ran) progress = true;
        // every lane of the wave is now waiting or done.  Lanes inside a lane-masked region (a divergent branch that
        // holds cross-lane operations, annotated in the sources) go first, as a partial wavefront -- the others wait at
        // the point where the branch rejoins, which is what the hardware's execution mask does.
        int depth = -1;
        for (unsigned t = t0; t < t1; ++t)
          if (w->lanes[t].state == WAIT_WAVE) depth = std::max(depth, w->lanes[t].depth);
        if (depth < 0) break;  // the wave is done or at a workgroup barrier
        const char* site = nullptr;
        uint64_t group = 0;
        for (unsigned t = t0; t < t1; ++t) {
          const Lane& l = w->lanes[t];
          if (l.state == DONE || l.depth != depth) continue;
          if (l.state == WAIT_BLOCK) die("lanes of one wavefront are at a cross-lane operation and at __syncthreads:", site, l.site);
          if (site && site != l.site && std::strcmp(site, l.site) != 0)
            die("lanes of one wavefront are at different cross-lane operations (a divergent region that needs "
                "PCLHIP_LANE_MASKED_REGION?):", site, l.site);
          site = l.site;
          group |= 1ull << (t - t0);
        }
        for (unsigned t = t0; t < t1; ++t)
          if ((group >> (t - t0)) & 1u) {
            Lane& l = w->lanes[t];
            w->waves[wv].val[t - t0] = l.pending;
            l.live = group;
            l.state = READY;
          }
        progress = true;
      }
    }
    if (done == nthreads) break;
    // all live lanes wait at the workgroup barrier
    bool all_block = true;
    for (unsigned t = 0; t < nthreads; ++t)
      if (w->lanes[t].state != DONE && w->lanes[t].state != WAIT_BLOCK) all_block = false;
    if (all_block) {
      for (unsigned t = 0; t < nthreads; ++t)
        if (w->lanes[t].state == WAIT_BLOCK) w->lanes[t].state = READY;
      progress = true;
    }
    if (!progress) die("deadlock inside a workgroup");
  }
  cur = nullptr;
}

// ---- pool --------------------------------------------------------------------------------------------------------------
struct Pool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  // the job
  uint64_t generation = 0;
  dim3 grid, block;
  const std::function<void()>* body = nullptr;
  std::atomic<unsigned> next{0};
  unsigned active = 0;
  bool quit = false;

  unsigned size() {
    if (const char* e = std::getenv("WAVESIM_THREADS")) return std::max(1, std::atoi(e));
    return std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  }
  void worker_main() {
    Worker me;
    tw = &me;
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
      }
      for (;;) {
        const unsigned b = next.fetch_add(1);
        if (b >= grid.x) break;
        run_block(&me, grid, block, b, *body);
      }
      {
        std::lock_guard<std::mutex> lk(m);
        if (--active == 0) cv_done.notify_all();
      }
    }
  }
  void run(dim3 g, dim3 b, const std::function<void()>& f) {
    std::unique_lock<std::mutex> lk(m);
    if (threads.empty()) {
      const unsigned n = size();
      for (unsigned i = 0; i < n; ++i) threads.emplace_back([this] { worker_main(); });
    }
    grid = g;
    block = b;
    body = &f;
    next.store(0);
    active = unsigned(threads.size());
    ++generation;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return active == 0; });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
  }
};
Pool& pool() {
  static Pool* p = new Pool;  // leaked on purpose: worker threads may outlive static destruction order otherwise
  return *p;
}
std::mutex launch_mutex;  // one launch at a time (streams are immediate)

}  // namespace

const uint64_t* exchange(uint64_t v, const char* site, uint64_t* live_mask) {
  Worker* w = tw;
  Lane* me = w->running;
  me->pending = v;
  me->site = site;
  me->state = WAIT_WAVE;
  yield_to_scheduler();
  *live_mask = me->live;
  return me->wave->val;
}

void masked_region_enter() { ++tw->running->depth; }
void masked_region_leave() { --tw->running->depth; }

void block_barrier(const char* site) {
  Worker* w = tw;
  Lane* me = w->running;
  me->site = site;
  me->state = WAIT_BLOCK;
  yield_to_scheduler();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  if (grid.x == 0 || block.x == 0) return;
  std::lock_guard<std::mutex> lk(launch_mutex);
  pool().run(grid, block, body);
}

}  // namespace wavesim

// ======================================================================================================================
// HIP runtime entry points (host memory, immediate streams)
// ======================================================================================================================
namespace {
std::mutex alloc_mutex;
std::map<uintptr_t, size_t> device_blocks;  // start -> bytes
thread_local hipError_t last_error = hipSuccess;
struct FakeEvent {
  std::chrono::steady_clock::time_point t;
};
int fake_stream_storage;
}  // namespace

extern "C" {

hipError_t hipMalloc(void** p, size_t bytes) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, bytes ? bytes : 256) != 0) return hipErrorOutOfMemory;
  std::memset(q, 0xCD, bytes < (size_t(1) << 20) ? bytes : (size_t(1) << 20));  // fresh device memory is not zero
  std::lock_guard<std::mutex> lk(alloc_mutex);
  device_blocks[reinterpret_cast<uintptr_t>(q)] = bytes ? bytes : 256;
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(alloc_mutex);
  device_blocks.erase(reinterpret_cast<uintptr_t>(p));
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  *p = std::calloc(1, bytes ? bytes : 16);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* f, size_t* t) {
  *f = size_t(8) << 30;
  *t = size_t(16) << 30;
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p) {
  std::lock_guard<std::mutex> lk(alloc_mutex);
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  auto it = device_blocks.upper_bound(a);
  if (it != device_blocks.begin()) {
    --it;
    if (a < it->first + it->second) {
      std::memset(attr, 0, sizeof *attr);
      attr->type = hipMemoryTypeDevice;
      attr->devicePointer = const_cast<void*>(p);
      return hipSuccess;
    }
  }
  last_error = hipErrorInvalidValue;
  return hipErrorInvalidValue;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  if (n) std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  if (n) std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                            hipStream_t) {
  for (size_t r = 0; r < height; ++r)
    std::memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) {
  if (n) std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  if (n) std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = reinterpret_cast<hipStream_t>(&fake_stream_storage);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipGetLastError() {
  const hipError_t e = last_error;
  last_error = hipSuccess;
  return e;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "wavesim: HIP error"; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* prop, int) {
  std::memset(prop, 0, sizeof *prop);
  std::snprintf(prop->name, sizeof prop->name, "wavesim (CPU emulation of a wavefront machine)");
  std::snprintf(prop->gcnArchName, sizeof prop->gcnArchName, "gfx950:wavesim");
  prop->multiProcessorCount = []() {
    const char* e = std::getenv("WAVESIM_CUS");
    return e ? std::max(1, std::atoi(e)) : 16;
  }();
  prop->warpSize = 64;
  prop->totalGlobalMem = size_t(16) << 30;
  prop->sharedMemPerBlock = 64 << 10;
  prop->maxThreadsPerBlock = 1024;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = reinterpret_cast<hipEvent_t>(new FakeEvent{std::chrono::steady_clock::now()});
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<FakeEvent*>(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  reinterpret_cast<FakeEvent*>(e)->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches are synchronous here
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(reinterpret_cast<FakeEvent*>(b)->t - reinterpret_cast<FakeEvent*>(a)->t).count();
  return hipSuccess;
}
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) {
  *n = 2;
  return hipSuccess;
}
hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) {
  std::memset(a, 0, sizeof *a);
  a->maxThreadsPerBlock = 1024;
  return hipSuccess;
}

}  // extern "C"

// ======================================================================================================================
// A stand-in for the collective library (icp_loop.hip binds these four under PCLHIP_WAVESIM instead of dlopen'ing RCCL):
// the ranks are PROCESSES on this host, the communicator is a POSIX shared-memory block named by the unique id, an
// all-reduce is "write my operand, barrier, add the ranks' operands in rank order, barrier".  It lets the N > 1 logic
// of the library itself -- communicator set-up from a shared id, the stream-ordered all-reduce between the reduction and
// the solve of every iteration, region masks with real peer processes -- run in the CPU tier.  It says nothing about
// RCCL or xGMI.
// ======================================================================================================================
#include <fcntl.h>
#include <sched.h>
#include <unistd.h>

namespace {
constexpr int NCCL_MAX_RANKS = 16, NCCL_MAX_COUNT = 2048;  // (the 2048-bin histograms of the rejectors' selections)
struct ShmComm {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  double slots[NCCL_MAX_RANKS][NCCL_MAX_COUNT];
};
struct FakeComm {
  ShmComm* shm = nullptr;
  int rank = 0, nranks = 1;
  char name[64];
};
struct FakeUniqueId {
  char internal[128];
};
bool shm_barrier(FakeComm* c) {
  ShmComm* s = c->shm;
  const uint32_t gen = s->generation.load(std::memory_order_acquire);
  if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == uint32_t(c->nranks)) {
    s->arrived.store(0, std::memory_order_relaxed);
    s->generation.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  for (uint64_t spins = 0; s->generation.load(std::memory_order_acquire) == gen; ++spins) {
    if (spins > 2000) usleep(50); else sched_yield();
    if (spins > 2400000) return false;  // ~2 minutes: a peer died
  }
  return true;
}
}  // namespace

extern "C" {

__attribute__((visibility("hidden"))) int ncclGetUniqueId(FakeUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  static std::atomic<unsigned> serial{0};
  std::snprintf(id->internal, sizeof id->internal, "/wavesim-nccl-%d-%u-%llx", int(getpid()), serial.fetch_add(1),
                (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}
__attribute__((visibility("hidden"))) int ncclCommInitRank(void** comm, int nranks, FakeUniqueId id, int rank) {
  if (nranks < 1 || nranks > NCCL_MAX_RANKS || rank < 0 || rank >= nranks) return 4;
  id.internal[63] = 0;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 2;
  if (ftruncate(fd, sizeof(ShmComm)) != 0) {
    close(fd);
    return 2;
  }
  void* p = mmap(nullptr, sizeof(ShmComm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  FakeComm* c = new FakeComm;
  c->shm = static_cast<ShmComm*>(p);  // a fresh object is all zeros: counters start at 0
  c->rank = rank;
  c->nranks = nranks;
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  if (!shm_barrier(c)) return 3;  // like ncclCommInitRank: returns when every rank has joined
  *comm = c;
  return 0;
}
__attribute__((visibility("hidden"))) int ncclCommDestroy(void* comm) {
  FakeComm* c = static_cast<FakeComm*>(comm);
  if (!c) return 0;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->shm, sizeof(ShmComm));
  delete c;
  return 0;
}
__attribute__((visibility("hidden"))) int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op,
                                                                 void* comm, hipStream_t) {
  FakeComm* c = static_cast<FakeComm*>(comm);
  const bool sum_f64 = dtype == 8 && op == 0, min_u64 = dtype == 5 && op == 3;  // (ncclFloat64, ncclSum) / (ncclUint64, ncclMin)
  if (!c || !(sum_f64 || min_u64)) return 4;
  // any count: NCCL_MAX_COUNT 8-byte elements at a time (write my operand, barrier, combine in rank order, barrier)
  for (size_t at = 0; at < count || (count == 0 && at == 0); at += NCCL_MAX_COUNT) {
    const size_t m = count - at < size_t(NCCL_MAX_COUNT) ? count - at : size_t(NCCL_MAX_COUNT);
    std::memcpy(c->shm->slots[c->rank], static_cast<const char*>(send) + at * 8, m * 8);
    if (!shm_barrier(c)) return 3;
    if (sum_f64) {
      double* out = static_cast<double*>(recv) + at;
      for (size_t i = 0; i < m; ++i) {
        double s = 0.0;
        for (int r = 0; r < c->nranks; ++r) s += c->shm->slots[r][i];
        out[i] = s;
      }
    } else {
      unsigned long long* out = static_cast<unsigned long long*>(recv) + at;
      for (size_t i = 0; i < m; ++i) {
        unsigned long long v = ~0ull;
        for (int r = 0; r < c->nranks; ++r) {
          unsigned long long x;
          std::memcpy(&x, &c->shm->slots[r][i], 8);
          v = x < v ? x : v;
        }
        out[i] = v;
      }
    }
    if (count == 0) break;
    if (at + NCCL_MAX_COUNT < count && !shm_barrier(c)) return 3;  // the next chunk overwrites the slots
  }
  if (!shm_barrier(c)) return 3;  // nobody overwrites its slot before everybody has read it
  return 0;
}

}  // extern "C"

// what pclhip_version() of a library linked with this runtime says (api.hip holds a weak reference to this function)
extern "C" __attribute__((visibility("default"))) const char* pclhip_emulation_banner(int verify_bounds) {
  return verify_bounds ? "pclhip 0.1 (wavesim: CPU emulation, test infrastructure; verify-bounds build)"
                       : "pclhip 0.1 (wavesim: CPU emulation, test infrastructure)";
}
