// Device contexts of libmarlin_hip.so: per context one GPU, its streams, cached twiddle table, scratch buffers, uploaded base sets,
// prover keys, shard configuration and HIP-event profiling.
#pragma once
#include <cstdlib>
#include <functional>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <vector>
#include <vector>
#include "../../include/marlin_hip.h"

namespace mh {
// MH_TRACE=2: a host-side timeline WITHOUT synchronisation -- where the calling thread's time goes between the HIP calls of a proof
// (the GPU is idle at every host round trip: a commit round's results, the Fiat-Shamir challenge, the next round's first launch).
// host_tick(label) records a time stamp; host_ticks_flush() prints the differences on stderr.
struct HostTicks { bool on = false; bool init = false; std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> t; };
inline HostTicks& host_ticks() { static thread_local HostTicks h; if (!h.init) { const char* e = getenv("MH_TRACE"); h.on = e && atoi(e) == 2; h.init = true; } return h; }
inline void host_tick(const char* label) { HostTicks& h = host_ticks(); if (h.on) h.t.emplace_back(label, std::chrono::steady_clock::now()); }
inline void host_ticks_flush() {
  HostTicks& h = host_ticks();
  if (!h.on) return;
  for (size_t i = 1; i < h.t.size(); i++)
    fprintf(stderr, "[mh_host] %9.1f us  %s -> %s\n", std::chrono::duration<double, std::micro>(h.t[i].second - h.t[i - 1].second).count(), h.t[i - 1].first, h.t[i].first);
  h.t.clear();
}


extern thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define MH_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t _e = (call);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _b[512];                                                                       \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return mh::fail(_e == hipErrorOutOfMemory ? MH_ENOMEM : MH_EHIP, _b);               \
    }                                                                                     \
  } while (0)

#define MH_TRY(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != MH_OK) return _r; \
  } while (0)

// Test hook (mh_debug_fail_scratch): the n-th Scratch::ensure call from now on fails as if the device were out of memory, whether
// or not it would have had to allocate -- how tests/test_gpu_rccl_native.py makes ONE rank of a sharded proof fail mid-prove.
extern int g_debug_poison_scratch;        // != 0: every fresh device allocation is filled with 0xA5 bytes (mh_debug_poison_scratch): a read of
                                          // memory nothing has written yet then returns garbage for sure instead of whatever the heap held
extern int g_debug_fail_scratch;          // > 0: calls left until the forced failure
extern int g_debug_corrupt;               // != 0: the next fixed-base MSM batch damages its own data after that stage, once (mh_debug_corrupt: the
                                          // fault injection that shows what MH_CHECK catches): 1 = one entry of the sorted lists, 2 = one bucket becomes garbage,
                                          // 3 = one bucket becomes a copy of its neighbour (a valid point), 4 = one plane becomes a copy of another
extern uint64_t g_debug_scratch_calls;    // Scratch::ensure calls so far (the test measures a proof with it)

// (the fill runs on the null stream, which the library's non-blocking streams do not wait for: drain the device on both sides, or
// the poison could land on top of what a kernel has already written there)
inline void debug_poison(void* p, size_t bytes) { (void)hipDeviceSynchronize(); (void)hipMemset(p, 0xA5, bytes); (void)hipDeviceSynchronize(); }

struct Scratch {
  void* ptr = nullptr;
  size_t cap = 0;
  bool hookable = true;                   // false: the staging buffers of the transport itself (a rank without them cannot enter a collective at all)
  int ensure(size_t bytes) {
    if (hookable) g_debug_scratch_calls++;
    if (hookable && g_debug_fail_scratch > 0 && --g_debug_fail_scratch == 0) return fail(MH_ENOMEM, "hipMalloc scratch failed (forced by mh_debug_fail_scratch)");
    if (bytes <= cap) return MH_OK;
    if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) {
      e = hipMalloc(&ptr, bytes);
      want = bytes;
      if (e != hipSuccess) { ptr = nullptr; return fail(MH_ENOMEM, "hipMalloc scratch failed"); }
    }
    cap = want;
    if (g_debug_poison_scratch) debug_poison(ptr, want);
    return MH_OK;
  }
  void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

// page-locked host staging for the small copies in a hot path (asynchronous for real, no bounce through the runtime's buffer)
struct Pinned {
  void* ptr = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return MH_OK;
    if (ptr) { (void)hipHostFree(ptr); ptr = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&ptr, want, hipHostMallocDefault) != hipSuccess) { ptr = nullptr; return fail(MH_ENOMEM, "hipHostMalloc staging failed"); }
    cap = want;
    return MH_OK;
  }
  void release() { if (ptr) (void)hipHostFree(ptr); ptr = nullptr; cap = 0; }
};

struct BaseSet {
  void* d_points = nullptr;  // G1Affine[n]
  size_t n = 0;
  // fixed-base window table (mh_bases_precompute): G1Affine[tab_W][n], level j = 2^{start_j} * points; level 0 is a copy
  void* d_table = nullptr;
  uint32_t tab_c = 0, tab_W = 0;
  bool tab_auto = false;     // width chosen by the library (re-chosen when the shard count changes)
};

struct G2Set { void* d_points = nullptr; size_t n = 0; };   // G2Affine[n] (x.c0 | x.c1 | y.c0 | y.c1, Montgomery)

// PF_MSM: wall time of the MSM groups on the main stream (accumulation included); PF_MSM_STAGES: the sort and bucket-reduction
// stages by themselves, on whichever stream they ran (they overlap the accumulation of the other sub-batch)
// PF_EXCHANGE: the collectives of a sharded proof (all-gather of partial points, all-to-all of the distributed transforms, all-gather of
// round polynomials), events on the library's stream around each
// PF_SIDE: whatever a side job (Context::side_job) runs on the second stream beside an MSM batch's bucket reduction -- kept apart from
// its own family so that the families of the main stream still add up to the step
// PF_MSM_REDUCE: the bucket reduction by itself (rsum + plane kernels; inside PF_MSM_STAGES) -- bench.py's roofline_reduce
enum ProfFamily { PF_NTT = 0, PF_MSM = 1, PF_MSM_ACCUM = 2, PF_GLUE = 3, PF_MSM_STAGES = 4, PF_EXCHANGE = 5, PF_SIDE = 6, PF_MSM_REDUCE = 7, PF_COUNT = 8 };

struct ProfRec { int family; hipEvent_t a, b; };

// What prover.hip keeps per context (prover keys, shard configuration, the job status of a sharded proof, the native transport's
// communicator): defined there, owned here so that it lives and dies with the context.
struct ContextExt { virtual ~ContextExt() = default; };

// One context = one GPU's worth of library state: streams, twiddles, workspaces, uploaded base sets, prover keys, shard
// configuration.  The process has a default context (mh_init); mh_ctx_create makes more -- on the same GPU (two proofs in flight from
// two threads) or on another one (one host process driving several GPUs: one thread per context) -- and mh_ctx_set_current binds the
// calling THREAD to one of them; every entry point works on the calling thread's current context and takes that context's lock, so
// calls into different contexts run side by side and calls into one context are serialised.
struct Context {
  bool inited = false;
  std::unique_ptr<ContextExt> ext;   // prover.hip: ProverState
  void* srs_table = nullptr;         // G1Affine[32][256]: the windowed multiples of the generator behind mh_srs_powers
  bool ntt_attr_done = false, msm_attr_done = false;   // hipFuncSetAttribute is per device: done once per context
  int device = -1;
  int num_simds = 1024;      // 4 per CU
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::recursive_mutex mu;

  // NTT
  void* tw = nullptr;        // Fr[2^tw_log]
  void* tw30 = nullptr;      // the same table as 9 x 30-bit limbs of w R' mod r (ntt30.cuh), 36 B per entry
  void* tw30s = nullptr;     // plain w and floor(w R' / r) for the Shoup twiddle products, 72 B per entry (ntt30.cuh: butterfly_shoup)
  uint32_t tw_log = 0;
  Scratch ntt_tmp[2];
  Scratch ntt_dist_buf[2], sl_send, sl_recv;
  std::map<uint64_t, Scratch> ntt_dist_tabs;   // twiddle hi / lo tables of the distributed transform, per (size, direction, rank, ranks)   // distributed transform (ntt_dist.cuh): exchange buffers, twiddle hi / lo tables
  Scratch io;                // staging for host-pointer entry points

  // MSM
  std::map<uint64_t, BaseSet> bases;
  std::map<uint64_t, G2Set> g2_bases;
  Scratch msm_dig, msm_sorted, msm_bh, msm_tot, msm_base, msm_buckets, msm_seg, msm_win, msm_pend;
  Scratch msm_gather;        // the bases of a skewed strided batch, gathered for the variable-base path
  // fixed-base path (msm_fb.cuh): the workspace of one job group
  struct FbWs {
    Scratch dig, val, sorted, pc, ptot, desc, blk, bh, tot, base, pend, buckets, seg, win, sums, perm, vtab, aux;
    Pinned h_desc, h_blk, h_out;              // host side of the descriptors (+ alias map), the block list, the results
    void release_all() {
      for (Scratch* b : {&dig, &val, &sorted, &pc, &ptot, &desc, &blk, &bh, &tot, &base, &pend, &buckets, &seg, &win, &sums, &perm, &vtab, &aux}) b->release();
      for (Pinned* b : {&h_desc, &h_blk, &h_out}) b->release();
    }
  } fbws;
  hipStream_t stream2 = nullptr;     // library-owned side stream: the caller's independent work beside a bucket reduction (side_job)
  // MH_DIAG (bit mask, diagnostics only -- tools/soak_sliced.py bisects with it): 2 = no host pool (the planes of every job are
  // combined on the calling thread).  (Bit 1 turned the sort's copy stream off in round 6's first soak; the one-pass sort has no
  // host round trip and hence no copy stream any more.)
  unsigned diag = 0;
  // Work of the CALLER that does not depend on the batch's results, issued on stream2 the moment the batch's accumulation has
  // been launched, behind an event recorded after it: it then runs beside the bucket reduction -- a latency chain at one wave per
  // SIMD that leaves a third of the VALU's issue slots idle -- and through the host round trip that follows.  The prover hands
  // over the three challenge-independent 4H transforms of round 2 this way during round 1's commitment (prover.hip).  The job is
  // called with `stream` temporarily set to stream2; side_ev[1] marks its end.  A batch that never reaches the fixed-base
  // accumulation (variable-base fallback) leaves the job untouched and the caller runs it itself.
  std::function<int()> side_job;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
  bool side_ran = false;
  uint64_t n_fb_groups = 0, n_vb_groups = 0;  // job groups that ran on the fixed-base / variable-base path
  Scratch tr_off[3], tr_cnt[2], tr_p[2], tr_sums, tr_ob, tr_pre, tr_prod, tr_scr;   // pair-tree accumulation

  // MH_CHECK / mh_check_level: the invariants of msm_check.cuh, checked per fixed-base batch.  `report` is the text of the last
  // checked batch (mh_check_report); a violation makes the MSM call return MH_ECHECK with the failing stage in mh_last_error
  struct Check {
    int level = 0;
    uint64_t batches = 0, violations = 0;
    std::string report;
    Scratch d; Pinned h;
  } chk;

  // profiling
  bool prof_on = false;
  unsigned prof_mask = ~0u;            // families whose scopes record events (mh_prof_enable)
  int prof_override = -1;              // >= 0: every scope opened meanwhile records under this family (side jobs: PF_SIDE)
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  double prof_ms[PF_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t prof_n[PF_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
};

Context& ctx();            // the calling thread's current context (mh_ctx_set_current), the default context otherwise
Context& default_ctx();
// true once a second context exists: entry points then select their context's device on the calling thread every time
extern bool g_multi_ctx;
extern std::atomic<uint64_t> g_next_handle;   // handles are unique across contexts
struct CtxLock {           // LOCKED_CTX: the context's lock, and its device current on this thread
  std::lock_guard<std::recursive_mutex> lk;
  explicit CtxLock(Context& c) : lk(c.mu) { if (g_multi_ctx && c.inited) (void)hipSetDevice(c.device); }
};

struct ProfScope {
  Context& c;
  int fam;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t st;
  ProfScope(Context& c_, int fam_, hipStream_t s_ = nullptr) : c(c_), fam(c_.prof_override >= 0 ? c_.prof_override : fam_), st(s_ ? s_ : c_.stream) {
    if (!c.prof_on || !((c.prof_mask >> fam) & 1u)) return;
    auto get = [&]() { hipEvent_t e; if (!c.ev_pool.empty()) { e = c.ev_pool.back(); c.ev_pool.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    a = get(); b = get();
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    c.prof.push_back({fam, a, b});
  }
};

}  // namespace mh
