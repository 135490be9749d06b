// pclhip_internal.hpp -- shared host/device definitions of the MI355X ICP hot path.
//
// Data layout in HBM (one pclhip_index):
//   pts   [n_pad]  float4   target points in kd order (index_build.hip); .w = original index (bit cast).
//                           n_pad = n rounded up to a whole leaf; pad slots hold +FLT_MAX
//                           sentinels with index 0xFFFFFFFF (distance overflows to +inf).
//   soa   [n1][4*LEAF] float the same points per leaf as x[16] y[16] z[16] w[16] (256 B): the block a
//                           wavefront stages in LDS with global_load_lds_dwordx4; candidate pairs
//                           then feed v_pk_* math from broadcast ds_read_b128.
//   nrm   [n_pad]  float4   (nx,ny,nz,curvature) in the same order (optional).
//   box[1][n1]     Box      tight AABB of every leaf = LEAF consecutive sorted points.
//   box[l][n_l]    Box      AABB of FANOUT consecutive boxes of level l-1 (implicit wide BVH:
//                           no pointers, child c of node i at level l is node i*FANOUT+c of l-1).
//   rank  [n_orig] uint32   original index -> sorted position (0xFFFFFFFF for dropped points).
// The top level has <= FANOUT boxes and is scanned by one wavefront (lane j <-> child j).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/pclhip.h"
#include "closed_forms.hpp"

namespace pclhip {

constexpr int LEAF = 16;        // points per leaf (candidates broadcast through SGPRs)
constexpr int FANOUT = 64;      // children per internal node = one per lane of a wavefront
constexpr int SCHED_CTR_STRIDE = 32;                       // uint32 between the counters of two XCDs (128 bytes)
constexpr size_t SCHED_CTR_BYTES = 8 * SCHED_CTR_STRIDE * 4;  // GroupFeed's counters of a context
constexpr int MAX_LEVELS = 6;   // levels 1..5: 16 * 64^5 points, far beyond the int32 indices of the API (2^31 points: 134M leaves ->
                                // 2.1M, 32768, 512, 8 boxes).  Every entry costs the kernels four scalar registers.
constexpr int WAVE = 64;
constexpr uint32_t NO_INDEX = 0xFFFFFFFFu;
constexpr int TOPCACHE_BOXES = 176;  // 5.5 KB of LDS per block

struct Box {       // 32 B: two aligned float4 loads
  float4 lo;       // xyz = min corner
  float4 hi;       // xyz = max corner
};

struct LevelInfo {  // one entry per tree level, read with a wave-uniform scalar load
  const Box* box;
  uint32_t count;
  uint32_t pad;
};

// Device-visible view of an index (passed to kernels by value).
struct IndexView {
  const float4* pts;
  const float* soa;             // per leaf: x[LEAF], y[LEAF], z[LEAF], w[LEAF] (LDS-staging copy)
  const float4* nrm;
  const float4* disc;           // per leaf two float4: (centre.xyz, R) (n.xyz, hn) of the bounded cylinder that holds
                                // the leaf's points (traverse.hpp: point_disc_lb).  nullptr: search with boxes only.
  float disc_from;              // discs are used where wave radius^2 > this (a few mean leaf diagonals, squared)
  const LevelInfo* lv;          // [MAX_LEVELS] in device memory; lv[1] = leaves
  const Box* box[MAX_LEVELS];   // the same table in kernel arguments (SGPRs): selected with a
  uint32_t count[MAX_LEVELS];   // wave-uniform switch, no memory access on the traversal's critical path
  const Box* topcache;          // boxes of levels >= cache_from, contiguous (<= TOPCACHE_BOXES): every
  uint32_t cache_off[MAX_LEVELS];  // block copies them to LDS once, so scans of the upper levels never
  int cache_from;               // leave the CU (MAX_LEVELS if nothing is cached)
  uint32_t cache_count;
  int top;                      // highest level (count[top] <= FANOUT)
  uint32_t n;                   // finite points
  uint32_t n_pad;
  uint32_t* sched_ctr;          // the context's group counters, one per XCD, 128 bytes apart (traverse.hpp: GroupFeed)
  // kd CELLS of the level-2 / level-3 nodes (LaneTree::qcell at quad levels 3 and 6; nullptr: none): a node's tight box
  // lies inside its cell, so the start-level test of traverse() passes on the cell wherever it passed on the box -- and
  // also where the box is thin (a flat piece of a surface: the box is as thick as the noise, the cell is unbounded)
  const Box* cell2;
  const Box* cell3;
};

// The per-lane search structure (lane_search.hpp; round 5): the same kd order seen as an implicit 4-ary tree over the
// leaves.  Level q holds cnt(q) = ceil(n_leaf / 4^q) nodes, node i of level q covers leaves [i 4^q, (i+1) 4^q); the
// levels lie one after the other in both arrays (level 0 first), so the offset of level q+1 is that of level q plus
// cnt(q) and a lane that moves one level at a time keeps it in a register.
//   qbox   tight AABB of every node (level 0 = the leaf boxes: pclhip_index::box[1] IS the front of this array)
//   qcell  a CELL of every node: an axis-aligned region (faces at +-inf where nothing bounds it) such that every point
//          of the index that does NOT belong to the node lies outside its interior (index_build.hip: quad_cell_kernel,
//          derived from the sibling boxes and checked while it is derived: a node whose siblings are not separated along
//          an axis gets an inverted cell that contains nothing).  A ball strictly inside the cell of a node therefore
//          holds no point of any other node: the search of that ball ends inside the node.
struct LaneTree {
  const Box* qbox;
  const Box* qcell;
  uint32_t nleaf;   // cnt(0)
  int top;          // the root's level: cnt(top) == 1
};
__host__ __device__ inline uint32_t lane_tree_count(uint32_t nleaf, int q) {
  return uint32_t((uint64_t(nleaf) + ((uint64_t(1) << (2 * q)) - 1u)) >> (2 * q));
}

// Axis-aligned region [lo, hi) of the rank that owns a query (target sharding, dist.hip): a source point
// takes part in an iteration only while its CURRENT position lies inside.  Unbounded sides are +-inf.
struct RegionBox {
  float lo[3], hi[3];
  int on;
};

// Device-resident state of the ICP loop (icp_loop.hip): the iteration is closed on the GPU
// (icp_solve_kernel), so consecutive iterations are queued back to back and the host only observes.
struct IcpControl {
  float T_apply[12];   // rows 0..2 of the transform the next search launch applies to the working cloud
  int restart;         // the next search launch starts an alignment: pristine source, no seeds
  int stop;            // the alignment is over: launches already queued behind it fall through
  int mode;            // PCLHIP_ICP_*
  int auto_restart;    // measurement loop: a finished alignment is followed by the next one, never stop
  int nr_iterations;   // of the running alignment
  int step;            // iterations completed since the loop was armed (index of the next step record)
  int log_capacity;
  int pad0;
  float guess[16];
  float final_T[16];
  float Tk[16];
  cf::Criteria crit;
  cf::CriteriaState st;
};

// Served groups under target sharding (search.hip: icp_own_* kernels).  A rank holds the whole source but serves only the
// points whose current position lies in its region; the source is kd-ordered, so whole 64-point groups lie outside.  In
// the device-driven loop every iteration first lists the groups whose (transformed) box touches the region, and the search
// and accumulation kernels walk that list: a rank touches ~ n / G source points per iteration instead of n.  A group that
// was skipped for some iterations is brought up to date on the fly from the transforms it missed (hist), in the order
// and arithmetic the working cloud would have seen -- bit for bit what the full pass computes.
#ifndef PCLHIP_OWN_HIST_CAP
#define PCLHIP_OWN_HIST_CAP 128  // (tests build with 3: alignments that outlast the history)
#endif
constexpr int OWN_HIST_CAP = PCLHIP_OWN_HIST_CAP;
struct OwnedState {
  uint32_t epoch;     // launch index inside the running alignment (0 = the launch that starts it)
  uint32_t overflow;  // more launches than OWN_HIST_CAP: every group is served from here on
  uint32_t pad[2];
  float hist[OWN_HIST_CAP][12];  // hist[e] = the transform launch e applies to the working cloud
};
struct OwnedGroups {  // kernel argument
  const uint32_t* list;      // served groups of this launch, ascending
  const uint32_t* count;     // their number
  uint32_t* stamp;           // per group: transforms applied to its working copy so far (0: none, read the pristine
                             // cloud); bit 31: its match entries are known to be empty
  const OwnedState* state;
};

// Device-resident state of one pass of the rejector chain (rejectors.hip): counts, ranks and thresholds never visit the
// host between the kernels of a chain.
struct RejState {
  unsigned int count;      // pairs alive in front of the last selecting rejector
  int mode;                // of that rejector: 0 nothing to filter, 1 threshold, 2 drop everything
  unsigned int trimmed;    // a Trimmed rejector of the chain cut the list (it then comes back ordered by distance)
  unsigned int pad;
  unsigned long long key;  // the selected key (distance bits, or (distance, query) for Trimmed)
  double median;           // MedianDistance: the median distance ...
  double threshold;        // ... times the factor
};

// One record per completed iteration, written by icp_solve_kernel into pinned host memory.
struct IcpStepRecord {
  int step;               // running index
  int iteration;          // nr_iterations_ after this iteration (0: it failed for lack of correspondences)
  int convergence_state;  // DefaultConvergenceCriteria state after hasConverged
  int converged;          // the alignment ended with converged_ = true
  int ended;              // the alignment ended with this iteration (either way)
  int similar;            // iterations_similar_transforms_
  double num_correspondences;
  double mse;
  double prev_mse;        // criteria memory after this iteration
  float Tk[16];           // transformation_ of this iteration
  float final_T[16];      // final_transformation_ after it
};

}  // namespace pclhip

struct pclhip_comm;

struct pclhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;
  int num_cus = 256;
  // reusable scratch
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  unsigned long long* stats = nullptr;  // 8 work counters (device), non-null when enabled
  uint32_t* sched_ctr = nullptr;        // 8 group counters (device), 128 bytes apart: the dynamic tail of GroupFeed
  std::mutex feed_mutex;                // keeps "zero the counters, launch" one step of the stream (PCLHIP_LAUNCH_FED)
  void* staging = nullptr;  // device staging for host inputs
  size_t staging_bytes = 0;
  // Device allocations of indices, registrations and temporaries are recycled through the context
  // (pclhip::dev_malloc / dev_free): hipMalloc + hipFree of the ~1 GB a 10M-point index build touches cost more
  // than the build's kernels.  All work of a context is ordered on its one stream, so a block handed out
  // again is never still in use.  Bounded by PCLHIP_CACHE_MB (default 16384); 0 turns the cache off.
  std::vector<std::pair<void*, size_t>> cache;   // free blocks (pointer, bytes)
  std::unordered_map<void*, size_t> live;         // blocks handed out
  size_t cached_bytes = 0;
  size_t cache_limit = size_t(16384) << 20;
  // pclhip_ctx_set_option (none of them changes a result)
  int opt_served_groups = 1;            // target sharding: the device-driven loop walks the served groups only
  int opt_lookahead = 1;                // pclhip_icp_align: iterations queued ahead of the host's knowledge
  long long opt_arena_mb = -1;          // automatic arena: -1 = 288 B per point of the first large cloud, 0 = none
  int opt_lane_search = 0;              // 1: seeded ICP launches one lane per query (lane.hip) -- exact, measured 2.5x SLOWER than the
                                        // wave-cooperative body at 10M points (profiles/r05_lane_search_ab.txt): kept as a tested option
  int opt_cell_start = 1;               // start-level test of the seeded descents on kd cells instead of tight boxes (A/B)
  float opt_standoff_thickness = 0.3f;  // launches without seeds: the stand-off search serves indices whose leaves are thinner than this
                                        // against their width (search.hip: the measurements behind the gate) ...
  int opt_standoff_max_mb = 1 << 24;    // ... and no larger than this (points + leaf blocks + boxes, 56 B per point): no limit by
                                        // default since round 6 (search.hip: the measurements)
  int opt_reseed = 1;                   // seeded search: one fresh seed for a group whose seeds are all far (A/B)
  int opt_lane_max_up = 2;              // ... quad levels the first pass climbs before it hands a query to the second
  float opt_lane_far = 0.25f;           // ... a seed beyond this many mean leaf diagonals (squared) is replaced by a descent
  std::mutex cache_mutex;
  // The QUERY entry points (pclhip_knn, pclhip_radius_search -- what PCL's `const` search virtuals call from OpenMP loops,
  // registration/include/pcl/registration/impl/correspondence_estimation.hpp:163-175, features/.../normal_3d_omp.hpp:76-81)
  // take this lock for their whole duration: concurrent callers on one context are served one after the other.
  std::recursive_mutex api_mutex;
  // Small pinned host blocks (control blocks, step rings, mirrored states of the registrations) are kept for the
  // context's lifetime: hipHostFree synchronises the device (170 us each, three per registration object).
  std::vector<std::pair<void*, size_t>> pinned_cache;  // free blocks (pointer, bytes)
  // One reservation in front of that cache (pclhip_ctx_reserve; made automatically for the first cloud of a million
  // points or more): allocations are carved out of it (first fit, freed ranges coalesce), so a context's FIRST index
  // build pays one hipMalloc instead of some forty.  What does not fit goes the way above.
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::map<size_t, size_t> arena_free;  // offset -> bytes of the free ranges
  bool arena_tried = false;             // the automatic reservation was attempted (once per context)
};

struct pclhip_index {
  pclhip_ctx* ctx = nullptr;
  uint64_t n_orig = 0;  // records in the user's cloud (index space of results)
  uint32_t n = 0;       // finite, selected points
  uint32_t n_pad = 0;
  float4* pts = nullptr;
  bool pts_borrowed = false;  // pts belongs to somebody else (build_index_over): not freed with the index
  float* soa = nullptr;
  float4* nrm = nullptr;
  float4* disc = nullptr;
  float leaf_diag2 = 0.0f;  // mean squared diagonal of the leaf boxes
  float disc_thickness = 1.0f;  // sum of the discs' half thicknesses / sum of their radii: how thin the leaves are
  uint32_t* rank = nullptr;
  pclhip::Box* box[pclhip::MAX_LEVELS] = {};
  // per-lane search structure (LaneTree above): box[1] is the front of qbox when it exists
  pclhip::Box* qbox = nullptr;
  pclhip::Box* qcell = nullptr;
  int qtop = 0;
  pclhip::LaneTree lane_tree() const;
  pclhip::LevelInfo* lv_dev = nullptr;
  pclhip::Box* topcache = nullptr;
  uint32_t cache_off[pclhip::MAX_LEVELS] = {};
  int cache_from = pclhip::MAX_LEVELS;
  uint32_t cache_count = 0;
  uint32_t count[pclhip::MAX_LEVELS] = {};
  int top = 0;
  float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};
  double build_ms = 0;
  double last_kernel_ms = 0;
  bool has_normals = false;
  bool scaled = false;              // built through a rescaling point representation: coordinates * scale
  float scale[3] = {1, 1, 1};
  pclhip::IndexView view() const;
};

struct pclhip_icp {
  pclhip_ctx* ctx = nullptr;
  pclhip_index* target = nullptr;
  uint64_t n_orig = 0;       // source records
  uint32_t n = 0;            // source points (all records; non-finite ones are flagged invalid)
  uint32_t n_finite = 0;     // ... of which finite: the front of the kd-ordered arrays
  // the staged device copy of a HOST source cloud's records (all fields), kept for pclhip_icp_transform_source
  void* src_records = nullptr;
  const void* src_records_host = nullptr;
  size_t src_records_stride = 0;
  uint64_t src_records_n = 0;
  float4* src_sorted0 = nullptr;   // Morton-ordered input (w = original index), pristine
  float4* src_cur = nullptr;       // working copy (input_transformed)
  float src_lo[3] = {0, 0, 0}, src_hi[3] = {0, 0, 0};   // bounding box of the finite source points as they were set
  float4* src_nrm_sorted0 = nullptr;  // source normals in the same order (symmetric objective), pristine
  float4* src_nrm_cur = nullptr;      // ... rotated along with the working copy
  bool enforce_same_direction_normals = true;  // icp.h:368
  uint32_t* match = nullptr;       // per sorted source slot: ORIGINAL target index or NO_INDEX
  uint32_t* match_pos = nullptr;   // ... and its sorted position (seed of the next iteration)
  float* match_d2 = nullptr;
  bool seeds_cleared = false;       // match_pos was just reset (pclhip_icp_reset): the next host-driven launch is a cold one
  double* partials = nullptr;      // [blocks][NSUMS]
  double* sums_dev = nullptr;      // [NSUMS]
  double* sums_host = nullptr;     // pinned
  int grid_blocks = 0;
  pclhip_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  // DefaultConvergenceCriteria state that persists across align() calls
  double prev_mse;
  int iterations_similar_transforms = 0;
  int convergence_state = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;  // ev_mid: end of the search kernel
  bool mid_recorded = false;
  double last_kernel_ms = 0;
  double last_search_ms = 0;  // the search kernel alone (two-kernel variant); = last_kernel_ms when fused
  double source_order_ms = 0; // GPU time of the spatial ordering in the last pclhip_icp_set_source
  // optional stages between search and accumulation
  std::vector<pclhip_rejector> rejectors;
  bool reciprocal = false;
  uint8_t* keep = nullptr;        // per sorted source slot: correspondence survives the chain
  double last_median = 0;
  int fetch_order = 0;            // 0 by query, 1 by (match, distance), 2 by distance
  // reciprocal correspondences: an index over the SOURCE that borrows src_cur as its point array (the working copy is in
  // kd order already) and is refitted to the moved cloud every iteration (rejectors.hip)
  pclhip_index* src_index = nullptr;
  bool trim_pending = false;      // a Trimmed rejector ran: fetch_order becomes 2 if rej_state_host->trimmed says it cut
  pclhip::RejState* rej_state = nullptr;       // device
  pclhip::RejState* rej_state_host = nullptr;  // pinned mirror, valid after a stream synchronisation
  // device-driven loop (icp_loop.hip)
  pclhip::IcpControl* ctl = nullptr;        // device
  pclhip::IcpControl* ctl_host = nullptr;   // pinned staging copy used to arm the loop
  pclhip::IcpStepRecord* steps = nullptr;   // pinned ring written by icp_solve_kernel
  int steps_capacity = 0;
  std::vector<hipEvent_t> step_events;      // 4 per ring slot: start, after search, after accumulate, after solve
  pclhip_comm* comm = nullptr;              // native RCCL all-reduce of the record (dist.hip); not owned
  const pclhip_comm* target_size_checked_for = nullptr;  // check_same_target_size passed with this communicator
  pclhip::RegionBox region = {{0, 0, 0}, {0, 0, 0}, 0};  // target sharding: the source points this rank serves
  // served-group lists of the device-driven loop under target sharding (one device block, made on first use)
  void* own_block = nullptr;
  uint32_t own_groups = 0;
  float4* own_gbox = nullptr;            // [2 * groups] box of every 64-point group of the pristine source
  uint32_t* own_stamp = nullptr;         // [groups]
  uint32_t* own_flags = nullptr;         // [groups] 1: served in this launch
  uint32_t* own_prefix = nullptr;        // [groups / 256 used] served groups per block of the flag kernel
  uint32_t* own_list = nullptr;          // [groups]
  uint32_t* own_tot = nullptr;           // [4] tot[0] = served groups
  uint2* own_partial = nullptr;          // scan scratch
  pclhip::OwnedState* own_state = nullptr;
  // per-lane seeded search (lane.hip): masks, block counts, the list of given-up queries and its length -- one device block
  void* lane_block = nullptr;
  uint32_t lane_cap = 0;
  unsigned long long* lane_mask = nullptr;
  uint32_t* lane_bcount = nullptr;
  uint32_t* lane_queue = nullptr;
  uint32_t* lane_tot = nullptr;
};

namespace pclhip {

// ---- error plumbing -----------------------------------------------------------------------
// Frees device allocations and destroys events on scope exit (every early error return included).
// context-cached device memory (api.hip): same contract as hipMalloc / hipFree
hipError_t dev_malloc(pclhip_ctx* ctx, void** p, size_t bytes);
hipError_t pinned_malloc(pclhip_ctx* ctx, void** p, size_t bytes);  // hipHostMalloc through the context's cache
void pinned_free(pclhip_ctx* ctx, void* p, size_t bytes);
template <class T>
inline hipError_t pinned_malloc(pclhip_ctx* ctx, T** p, size_t bytes) {
  return pinned_malloc(ctx, reinterpret_cast<void**>(p), bytes);
}
void dev_free(pclhip_ctx* ctx, void* p);
void dev_cache_release(pclhip_ctx* ctx);  // really frees every cached block
pclhip_status reserve_arena(pclhip_ctx* ctx, size_t bytes);
// reserve the context's arena for clouds of `points` points (no-op when one exists or PCLHIP_ARENA_MB=0)
void dev_reserve_for_points(pclhip_ctx* ctx, uint64_t points);
// load the code objects of every translation unit now (hipFuncGetAttributes of one kernel each), not inside the first
// timed launch
void preload_code_objects(pclhip_ctx* ctx);
void preload_search_kernels(pclhip_ctx* ctx);
void preload_index_build_kernels();
void preload_voxelgrid_kernels();
void preload_rejector_kernels();
void preload_radius_kernels();
void preload_lane_kernels();
bool search_built_with_verify_bounds();  // search.hip compiled with -DPCLHIP_VERIFY_BOUNDS (traverse.hpp)
// lane.hip: the seeded search launches of an iteration, one lane per query (needs the target's LaneTree)
bool lane_search_available(const pclhip_icp* icp);
pclhip_status launch_lane_search(pclhip_icp* icp, const float T12[12], const pclhip::IcpControl* ctl, int order, float bound,
                                 bool use_max);
template <class T>
inline hipError_t dev_malloc(pclhip_ctx* ctx, T** p, size_t bytes) {
  return dev_malloc(ctx, reinterpret_cast<void**>(p), bytes);
}

struct DeviceScope {
  pclhip_ctx* ctx = nullptr;
  std::vector<void*> mem;
  std::vector<hipEvent_t> events;
  DeviceScope() = default;
  explicit DeviceScope(pclhip_ctx* c) : ctx(c) {}
  ~DeviceScope() {
    for (void* p : mem)
      if (p) dev_free(ctx, p);
    for (hipEvent_t e : events)
      if (e) (void)hipEventDestroy(e);
  }
  template <class T>
  hipError_t alloc(T** ptr, size_t bytes) {
    *ptr = nullptr;
    const hipError_t e = dev_malloc(ctx, reinterpret_cast<void**>(ptr), bytes ? bytes : 16);
    if (e == hipSuccess) mem.push_back(*ptr);
    return e;
  }
  hipError_t event(hipEvent_t* ev) {
    *ev = nullptr;
    const hipError_t e = hipEventCreate(ev);
    if (e == hipSuccess) events.push_back(*ev);
    return e;
  }
};

void set_error(pclhip_ctx* ctx, const std::string& msg);
#define PCLHIP_CHECK_HIP(ctx, expr)                                                         \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      ::pclhip::set_error((ctx), std::string(#expr) + ": " + hipGetErrorString(e__));       \
      return PCLHIP_ERR_HIP;                                                                \
    }                                                                                       \
  } while (0)
#define PCLHIP_REQUIRE(ctx, cond, msg)                     \
  do {                                                     \
    if (!(cond)) {                                         \
      ::pclhip::set_error((ctx), (msg));                   \
      return PCLHIP_ERR_INVALID;                           \
    }                                                      \
  } while (0)

// Resolve a user pointer: returns a device pointer valid on ctx->stream.  Host memory is staged
// (async copy from pageable memory is synchronous w.r.t. the host, which is what we want).
pclhip_status to_device(pclhip_ctx* ctx, const void* p, size_t bytes, const void** dev, void** owned);
bool is_device_pointer(const void* p);
pclhip_status ensure_scratch(pclhip_ctx* ctx, size_t bytes);

// ---- build steps (index_build.hip) ----------------------------------------------------------
// Sort `n` records (strided, device) into kd order (index_build.hip).  Outputs: sorted float4 (w = original index),
// number of finite points, bbox.  `sel` optionally selects a subset (device int32).
pclhip_status kd_order(pclhip_ctx* ctx, const void* dev_points, size_t stride, uint64_t n_records,
                       const int32_t* dev_sel, uint64_t n_sel, float4* out_sorted, uint32_t out_capacity,
                       uint32_t* out_n_finite, float lo[3], float hi[3], bool keep_nonfinite_at_end,
                       uint32_t* rank_or_null, bool ids_from_w = false, const float* scale = nullptr);
// the order of indices and sources alike: kd_order without ids taken from w
pclhip_status spatial_order(pclhip_ctx* ctx, const void* dev_points, size_t stride, uint64_t n_records,
                            const int32_t* dev_sel, uint64_t n_sel, float4* out_sorted, uint32_t out_capacity,
                            uint32_t* out_n_finite, float lo[3], float hi[3], bool keep_nonfinite_at_end,
                            uint32_t* rank_or_null, const float* scale = nullptr);
// with_discs also says "a kd index of its own points" (not the borrowed, refitted index of the reciprocal test): only such
// an index gets the per-lane search structure (cells need the kd partition)
pclhip_status build_boxes(pclhip_index* ix, bool with_discs = true);
pclhip_status build_index_over(pclhip_ctx* ctx, float4* pts_in_kd_order, uint32_t n_finite, uint32_t n_orig, pclhip_index** out);
// shard_dev.hip: partition / halo selection of a cloud in device memory (shard.cpp's results, bit for bit)
pclhip_status partition_slabs_device(const void* points, size_t stride, uint64_t n, int n_slabs, float* regions);
pclhip_status select_region_device(const void* points, size_t stride, uint64_t n, const float lo[3], const float hi[3],
                                   int32_t* out_indices, uint64_t capacity, uint64_t* out_count);
pclhip_status refit_boxes(pclhip_index* ix);  // the points of a built index moved: boxes again, stream-ordered
int icp_grid_blocks(pclhip_ctx* ctx, uint32_t ns);
pclhip_status launch_estimate_pairs(pclhip_ctx* ctx, int mode, const float4* src, const float4* src_nrm,
                                    const float4* tgt, const float4* tgt_nrm, const float* weights, uint32_t n, bool enforce,
                                    double* sums);
pclhip_status launch_normals_radius(pclhip_index* ix, double radius, const float vp[3], uint64_t* nan_count);
// normals at arbitrary query points (Feature::setSearchSurface): `queries` dense float4 in slot order; the radius form
// takes them in kd order with w = slot
pclhip_status launch_normals_at(pclhip_index* ix, const float4* queries, uint32_t nq, int k, double radius, const float vp[3],
                                float4* out, uint64_t* nan_count);
pclhip_status launch_normals_radius_at(pclhip_index* ix, const float4* queries_sorted, uint32_t nq, double radius,
                                       const float vp[3], float4* out, uint64_t* nan_count);
pclhip_status launch_gicp_covariances(pclhip_index* ix, int k, double eps, double* cov_sorted);
pclhip_status launch_fitness_score(pclhip_icp* icp, const float T[16], double max_range, double* score,
                                   uint64_t* nr);
// icp_loop.hip
pclhip_status icp_align_device(pclhip_icp* icp, const pclhip_icp_params* params, const float* guess,
                               pclhip_icp_result* res);
bool icp_is_sharded(const pclhip_icp* icp);
// under sharding only per-pair filters (the Distance rejector) are allowed: PCLHIP_ERR_STATE otherwise
pclhip_status sharded_filters_ok(pclhip_icp* icp);

// A divergent branch that holds cross-lane operations (ballots among the lanes that took it) is marked with this for the
// CPU emulation of the wavefront in the test tier (tests/wavesim/wavesim.hpp, which defines it); on the device the
// execution mask does the same and the marker is nothing.
#ifndef PCLHIP_LANE_MASKED_REGION
#define PCLHIP_LANE_MASKED_REGION (void)0
#endif

// Kernels that take part of their groups from the context's counters (traverse.hpp: GroupFeed) are launched through
// this: the counters are zeroed in stream order first, and no other thread's launch gets between the two.
#define PCLHIP_LAUNCH_FED(ctx, ...)                                                                \
  do {                                                                                             \
    std::lock_guard<std::mutex> pclhip_feed_lock((ctx)->feed_mutex);                               \
    (void)hipMemsetAsync((ctx)->sched_ctr, 0, pclhip::SCHED_CTR_BYTES, (ctx)->stream);             \
    hipLaunchKernelGGL(__VA_ARGS__);                                                               \
  } while (0)

// ---- kernels launched from api.cpp ----------------------------------------------------------
// timed = false: no events, no wait (k <= 32): the launch is queued and the call returns
// batches that are sparse against the index: fewer queries per wavefront (api.hip)
uint32_t sparse_fill(uint64_t nq, uint64_t n_index);
pclhip_status sparse_layout(pclhip_ctx* ctx, const float4* q_sorted, uint64_t nq, uint64_t n_index, float4** out, uint32_t* n_out);
pclhip_status launch_knn(pclhip_index* ix, const float4* q_sorted, uint32_t nq, int k,
                         int32_t* out_idx_sorted, float* out_d2_sorted, bool timed = true);
pclhip_status launch_normals(pclhip_index* ix, int k, const float vp[3], uint64_t* nan_count);
pclhip_status launch_icp_iterate(pclhip_icp* icp, const float T[16], float max_d2, bool use_max,
                                 int mode, hipEvent_t* step_events = nullptr);
// target sharding, after a run of the device-driven loop: the working copies of the groups the last launches did not
// serve are brought up to date (search.hip: served groups), stream-ordered
pclhip_status owned_groups_catch_up(pclhip_icp* icp, int mode);
// sums icp->sums_dev over the ranks on the context's stream (native RCCL communicator or the hook); no-op
// for a single-GPU registration
pclhip_status allreduce_record(pclhip_icp* icp);
pclhip_status allreduce_doubles(pclhip_icp* icp, double* device_buf, int count);  // any buffer (rejector histograms)
pclhip_status allreduce_min_u64(pclhip_icp* icp, unsigned long long* device_buf, size_t count);  // native communicator only
pclhip_status check_same_target_size(pclhip_icp* icp);  // OneToOne under target sharding: same n_orig on every rank
// The reciprocal test as one seeded search (search.hip): slot i asks the SOURCE index for the nearest neighbour of its
// matched target point, seeded by source point i itself (the index's positions are the source's slots), and drops the pair unless that is
// the answer (impl/correspondence_estimation.hpp:247-270).  Stream-ordered, no wait.
pclhip_status launch_recip_search(pclhip_index* src_ix, const float4* tgt_pts, const uint32_t* match_pos,
                                  const float4* cur, uint32_t n, float max_d2, bool use_max,
                                  uint8_t* keep);
// rejectors.hip: reciprocal filter + rejector chain on icp->keep (stream-ordered, may synchronise)
pclhip_status apply_correspondence_filters(pclhip_icp* icp, float max_d2, bool use_max);
// the collectives of the chain's selection passes for a rank whose source share is empty (zero histograms, same order)
pclhip_status apply_empty_shard_collectives(pclhip_icp* icp);
// kd order of float4 records whose .w already holds the point's id (used for the reciprocal index)
pclhip_status build_index_from_float4(pclhip_ctx* ctx, const float4* dev_pts_with_ids, uint32_t n, pclhip_index** out);

// ---- host closed forms (host_math.cpp) -------------------------------------------------------
void solve_point_to_plane(const double* sums, float* T);
void solve_point_to_point(const double* sums, float* T);
void solve_symmetric(const double* sums, float* T);
void mat4_mul_f32(const float* A, const float* B, float* C);

}  // namespace pclhip
