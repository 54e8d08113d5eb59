// Marlin index + prove with every polynomial resident in HBM.
//
// Host-side mirror (C++, because no Rust toolchain exists in this image) of the reference's
//   Marlin::index   /root/reference src/lib.rs:100-148  (+ src/ahp/indexer.rs:151-234,
//                   src/ahp/constraint_systems.rs:125-262)
//   Marlin::prove   /root/reference src/lib.rs:151-311  (+ src/ahp/prover.rs:211-706,
//                   src/ahp/verifier.rs:44-188, src/ahp/mod.rs:110-221, src/rng.rs:54-79)
// with PC = MarlinKZG10<Bls12_381> and FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng>
// (src/test.rs:128-130).  Function names follow the reference.  The transcript, the
// challenges and the 3-coefficient hiding polynomials live on the host; every O(n) object lives
// on the device and is only ever touched by the kernels in ntt.cuh / msm.cuh / poly.cuh.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <memory>
#include "drivers.h"
#include "ff.cuh"
#include "fs_host.h"
#include "g1.cuh"
#include "host_ff.h"
#include "host_pool.h"
#include "poly.cuh"
#include "ntt_dist.cuh"
#include "rng.cuh"
#include "wire_host.h"
#include "verify_host.h"
#include "rccl_native.h"
#include <chrono>
#include <cstdlib>

using namespace mh;
using hostff::HFq;
using hostff::HFr;
using hostff::HG1;
using hostff::HG1Affine;
using poly::FrArg;
using hostff::FQ_L; using hostff::FQ_B; using hostff::PT_B; using hostff::AFF_L; using hostff::XYZ_L;

namespace {

inline Fr dfr(const HFr& h) { Fr r; memcpy(r.v, h.v, 32); return r; }
inline FrArg arg(const HFr& h) { FrArg a; a.v = dfr(h); return a; }
inline uint32_t log2u(uint64_t n) { uint32_t l = 0; while ((1ull << l) < n) l++; return l; }
inline uint64_t np2(uint64_t n) { return 1ull << log2u(n < 1 ? 1 : n); }

// device allocation owned by a prover key or a scope: released on destruction, so that a failing index / prove does not
// leak what it had allocated so far (at process exit the runtime may already be gone: hipFree's error is ignored)
struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  ~DBuf() { release(); }
  int alloc(size_t b) {
    release();
    if (b == 0) b = 32;
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) { p = nullptr; return fail(MH_ENOMEM, "hipMalloc failed in prover key allocation"); }
    if (g_debug_poison_scratch) debug_poison(p, b);
    bytes = b;
    return MH_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  Fr* fr() const { return (Fr*)p; }
};

struct Csr { DBuf row_ptr, col, val; bool has_val = false; uint64_t nnz = 0; };

struct HidingRand { std::vector<HFr> blind; };          // kzg10::Randomness (blinding polynomial coeffs)
struct PolyRand { HidingRand rand; bool has_shifted = false; HidingRand shifted; };   // marlin_pc::Randomness

struct ProverKey {
  uint64_t nc = 0, ni = 0, nnz = 0;         // constraints (= variables), formatted inputs, non-zeros of the joint matrix
  uint64_t H = 0, K = 0, X = 0;
  uint32_t logH = 0, logK = 0, logX = 0;
  uint64_t srs_g = 0, srs_max_degree = 0, index_max_degree = 0;
  int pc = 0;                      // 0 = MarlinKZG10 (src/test.rs:123), 1 = SonicKZG10 (benches/bench.rs:81)
  HG1Affine gamma_g[3];            // powers_of_gamma_g[0..3)
  HG1Affine gamma_g_h[3], gamma_g_k[3];   // Sonic: shifted_powers_of_gamma_g for the bounds |H|-2 and |K|-2
  // index (device)
  DBuf ev_row, ev_col, ev_row_col, ev_val_a, ev_val_b, ev_val_c;       // evals on K
  DBuf p_row, p_col, p_a_val, p_b_val, p_c_val, p_row_col;            // coefficient form (K each)
  DBuf cs_row, cs_col, cs_row_col, cs_val_a, cs_val_b, cs_val_c;       // evals on the coset g K (third round)
  DBuf gpow_hi, gpow_lo, ginv_hi, ginv_lo;                             // g^(+-i) = hi[i >> 11] * lo[i & 2047]
  HFr coset_g;                                                         // g, with g^K != 1
  Csr A, B;
  DBuf t_items, t_erow, t_ecoef, t_item_ptr, t_group_ptr;
  uint64_t t_nitems = 0, t_ngroups = 0; bool t_has_coef = false;
  std::vector<fsh::Commitment> index_comms;
  std::vector<uint8_t> vk_bytes;
  // persistent polynomials of one proof + scratch
  DBuf z, za_ev, zb_ev, xpoly, w, za, zb, mask, t, g1, h1, g2, h2, outer, inner;
  DBuf S[8];
  DBuf small;       // partial sums / carries
  DBuf scal;        // a few device scalars (scan totals)
  // sliced sections (multi-GPU): this rank's M-layout blocks of the 12 index evaluation vectors and its twist
  // tables g^(+-(r + G j)), built when the shard geometry is first seen
  DBuf sl_ev[6], sl_cs[6], sl_gpow_hi, sl_gpow_lo, sl_ginv_hi, sl_ginv_lo;
  int sl_rank = -1, sl_world = 0;
  std::map<std::string, std::pair<const Fr*, uint64_t>> last_polys;   // prover oracles of the last proof (label -> ptr, len)
  void free_all() {
    DBuf* all[] = {&ev_row, &ev_col, &ev_row_col, &ev_val_a, &ev_val_b, &ev_val_c, &p_row, &p_col, &p_a_val, &p_b_val, &p_c_val,
                   &p_row_col, &cs_row, &cs_col, &cs_row_col, &cs_val_a, &cs_val_b, &cs_val_c, &gpow_hi, &gpow_lo, &ginv_hi, &ginv_lo,
                   &A.row_ptr, &A.col, &A.val, &B.row_ptr, &B.col, &B.val, &t_items, &t_erow, &t_ecoef, &t_item_ptr, &t_group_ptr,
                   &z, &za_ev, &zb_ev, &xpoly, &w, &za, &zb, &mask, &t, &g1, &h1, &g2, &h2, &outer, &inner, &small, &scal};
    for (auto* b : all) b->release();
    for (auto& s : S) s.release();
    for (auto& b : sl_ev) b.release();
    for (auto& b : sl_cs) b.release();
    for (DBuf* b : {&sl_gpow_hi, &sl_gpow_lo, &sl_ginv_hi, &sl_ginv_lo}) b->release();
    sl_rank = -1; sl_world = 0;
  }
};


// ---- small launch helpers ------------------------------------------------------------------
#define KLAUNCH(kern, n, ...)                                                                          \
  do {                                                                                                 \
    hipLaunchKernelGGL(kern, dim3(poly::grid_for(n)), dim3(poly::TPB), 0, c.stream, __VA_ARGS__);      \
  } while (0)

int zero_tail(Context& c, Fr* p, uint64_t from, uint64_t to) {
  if (to > from) MH_HIP(hipMemsetAsync(p + from, 0, (to - from) * 32, c.stream));
  return MH_OK;
}
int d2d(Context& c, Fr* dst, const Fr* src, uint64_t n) {
  if (n) MH_HIP(hipMemcpyAsync(dst, src, n * 32, hipMemcpyDeviceToDevice, c.stream));
  return MH_OK;
}
int h2d(Context& c, void* dst, const void* src, size_t bytes) {
  if (bytes) MH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c.stream));
  return MH_OK;
}
int set_fr(Context& c, Fr* dst, const HFr& v) {     // synchronous-safe: value copied from pageable host memory
  MH_HIP(hipMemcpyAsync(dst, v.v, 32, hipMemcpyHostToDevice, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}
int get_fr(Context& c, HFr* out, const Fr* src) {
  MH_HIP(hipMemcpyAsync(out->v, src, 32, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

// scratch: level l parks its n_l prefix products and hands n_{l+1} = ceil(n_l / 8192) * 256 thread totals to the next
// level, which again needs n_{l+1} prefix slots, until n_l <= 512 (host): n + 2 (n_1 + n_2 + ...) + 256 elements, i.e.
// at most n + n / 15 + 3 * 256 (n = 2^20: n + 2 * 32768 + 2 * 1024 + 256 = n + 67840).  Callers pass max(2K, 4H) + 64.
// ark_ff::batch_inversion (prover.rs:663, mod.rs:314) of n device elements, zeros left untouched, every inverse optionally
// multiplied by `scale`.  Montgomery's trick in levels: a level's threads multiply up 32 elements each (binv_fwd), the
// per-thread products are inverted by the next level, and a backward sweep applies them (binv_bwd).  The ONE inversion
// everything waits for is done on the host: a Fermat ladder on the device is ~380 dependent multiplications (0.3 ms
// for a few thousand threads, whatever their number), the host does it in ~30 us behind a 2 KB copy.
static int batch_inverse_level(Context& c, Fr* data, Fr* scratch, uint64_t n, const HFr& scale, int do_scale) {
  const uint64_t tile = (uint64_t)poly::TPB * poly::INV_CH;
  if (n <= 512) {
    // bottom: host
    std::vector<uint64_t> h(4 * n);
    MH_HIP(hipMemcpyAsync(h.data(), data, n * 32, hipMemcpyDeviceToHost, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));
    std::vector<HFr> v(n), pre(n);
    HFr acc = HFr::one();
    for (uint64_t i = 0; i < n; i++) { memcpy(v[i].v, &h[4 * i], 32); pre[i] = acc; if (!v[i].is_zero()) acc = acc * v[i]; }
    HFr inv = acc.inv();
    if (do_scale) inv = inv * scale;             // the scale rides on the running inverse: r_i = scale / v_i
    for (uint64_t i = n; i-- > 0;) {
      if (v[i].is_zero()) continue;
      HFr r = inv * pre[i];
      inv = inv * v[i];
      memcpy(&h[4 * i], r.v, 32);
    }
    MH_HIP(hipMemcpyAsync(data, h.data(), n * 32, hipMemcpyHostToDevice, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));       // h goes out of scope
    return MH_OK;
  }
  const uint64_t nblk = (n + tile - 1) / tile, nt = nblk * poly::TPB;
  Fr* totals = scratch + n;
  Fr* scratch2 = totals + nt;
  hipLaunchKernelGGL(poly::binv_fwd_kernel, dim3((unsigned)nblk), dim3(poly::TPB), 0, c.stream, (const Fr*)data, scratch, totals, (u64)n);
  MH_TRY(batch_inverse_level(c, totals, scratch2, nt, HFr::one(), 0));
  hipLaunchKernelGGL(poly::binv_bwd_kernel, dim3((unsigned)nblk), dim3(poly::TPB), 0, c.stream, data, (const Fr*)scratch, (const Fr*)totals,
                     (u64)n, arg(scale), do_scale);
  MH_HIP(hipGetLastError());
  return MH_OK;
}
int batch_inverse(Context& c, Fr* data, Fr* scratch, uint64_t n, const HFr& scale, int do_scale) {
  ProfScope ps(c, PF_GLUE);
  if (n == 0) return MH_OK;
  return batch_inverse_level(c, data, scratch, n, scale, do_scale);
}

struct Term { const Fr* p; uint64_t len; HFr coef; };
int lincomb(Context& c, Fr* out, uint64_t n, const std::vector<Term>& terms) {
  poly::LinComb lc;
  if (terms.size() > (size_t)poly::MAX_TERMS) return fail(MH_EINVAL, "lincomb: too many terms");
  lc.nterms = (int)terms.size();
  for (int j = 0; j < lc.nterms; j++) {
    lc.src[j] = terms[j].p; lc.len[j] = terms[j].len; lc.coef[j] = dfr(terms[j].coef);
    lc.is_one[j] = terms[j].coef == HFr::one();
  }
  ProfScope ps(c, PF_GLUE);
  KLAUNCH(poly::lincomb_kernel, n, out, (u64)n, lc);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// p(z) for a device polynomial.  Long polynomials first collapse 16-coefficient chunks by Horner (one multiplication per
// coefficient, no powers), then the chunk values are evaluated at z^16 by the block kernel (whose per-thread z^(start)
// costs ~2 multiplications per coefficient it covers -- affordable on 1/16 of the data).
// d_slot != nullptr: the value is left in device memory there (no host round trip; *out is not written)
int eval_poly(Context& c, ProverKey& pk, const Fr* p, uint64_t len, const HFr& z, HFr* out, Fr* d_slot = nullptr) {
  if (len == 0) {
    if (d_slot) { MH_HIP(hipMemsetAsync(d_slot, 0, 32, c.stream)); return MH_OK; }
    *out = HFr::zero(); return MH_OK;
  }
  Fr* base = pk.small.fr();
  HFr zz = z;
  {
    ProfScope ps(c, PF_GLUE);
    if (len >= (1u << 16)) {
      const uint64_t nch = (len + poly::LIN_CH - 1) / poly::LIN_CH;
      KLAUNCH(poly::divlin_chunk_kernel, nch, base, p, (u64)len, arg(z));
      for (int e = poly::LIN_CH; e > 1; e >>= 1) zz = zz.sqr();
      p = base; len = nch; base += nch;
    }
    uint64_t nthreads = (len + poly::EV_CH - 1) / poly::EV_CH;
    unsigned blocks = poly::grid_for(nthreads);
    Fr* partial = base;
    hipLaunchKernelGGL(poly::eval_partial_kernel, dim3(blocks), dim3(poly::TPB), 0, c.stream, partial, p, (u64)len, arg(zz));
    hipLaunchKernelGGL(poly::sum_kernel, dim3(1), dim3(poly::TPB), 0, c.stream, partial + blocks, (const Fr*)partial, (u64)blocks);
    base = partial + blocks;
  }
  MH_HIP(hipGetLastError());
  if (d_slot) return d2d(c, d_slot, base, 1);
  return get_fr(c, out, base);
}

// (q, -) = p / (X^n - 1); q gets len - n coefficients.  scratch: nchunks * n elements.
int div_vanishing(Context& c, Fr* q, const Fr* p, uint64_t len, uint64_t n, Fr* scratch) {
  if (len <= n) return MH_OK;
  uint64_t nrows = (len + n - 1) / n;
  uint64_t nchunks = (nrows - 1 + poly::DIV_ROWS - 1) / poly::DIV_ROWS;
  ProfScope ps(c, PF_GLUE);
  KLAUNCH(poly::divvan_partial_kernel, n * nchunks, scratch, p, (u64)len, (u64)n, (u64)nchunks);
  if (nchunks > 64 && n <= 65536)
    hipLaunchKernelGGL(poly::divvan_scan_kernel, dim3((unsigned)n), dim3(poly::TPB), 0, c.stream, scratch, (u64)n, (u64)nchunks);
  else
    KLAUNCH(poly::divvan_scan_simple_kernel, n, scratch, (u64)n, (u64)nchunks);
  KLAUNCH(poly::divvan_final_kernel, n * nchunks, q, (const Fr*)scratch, p, (u64)len, (u64)n, (u64)nchunks);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// q = (p - p(z)) / (X - z); q gets len - 1 coefficients.  scratch >= 2 * (len/16 + len/256 + ...) elements
int div_linear(Context& c, Fr* q, const Fr* p, uint64_t len, const HFr& z, Fr* scratch) {
  if (len <= 1) return MH_OK;
  ProfScope ps(c, PF_GLUE);
  // levels: A_0 = p; A_{j+1}[c] = Horner(A_j chunk c, m_j), m_0 = z, m_{j+1} = m_j^LIN_CH
  std::vector<const Fr*> A; std::vector<uint64_t> L; std::vector<HFr> M;
  A.push_back(p); L.push_back(len); M.push_back(z);
  Fr* cur = scratch;
  std::vector<Fr*> V;      // V[j] = A_{j+1}
  while (true) {
    uint64_t n = L.back();
    uint64_t nch = (n + poly::LIN_CH - 1) / poly::LIN_CH;
    Fr* v = cur; cur += nch;
    KLAUNCH(poly::divlin_chunk_kernel, nch, v, A.back(), (u64)n, arg(M.back()));
    V.push_back(v);
    HFr m = M.back();
    for (int e = poly::LIN_CH; e > 1; e >>= 1) m = m.sqr();     // ^LIN_CH (a power of two)
    A.push_back(v); L.push_back(nch); M.push_back(m);
    if (nch <= (uint64_t)poly::LIN_CH) break;
  }
  // top: carries into the chunks of the last A_j (j = A.size()-2) from A_{j+1} = V.back()
  int top = (int)V.size() - 1;
  std::vector<Fr*> C(V.size());
  for (size_t j = 0; j < V.size(); j++) { C[j] = cur; cur += L[j + 1]; }
  hipLaunchKernelGGL(poly::divlin_top_kernel, dim3(1), dim3(1), 0, c.stream, C[top], (const Fr*)V[top], (u64)L[top + 1], arg(M[top + 1]));
  for (int j = top - 1; j >= 0; j--) {
    // carries into chunks of A_j (count L[j+1]) from group carries C[j+1] (groups of LIN_CH chunks), values V[j], multiplier M[j+1]
    uint64_t ngroups = L[j + 2];
    KLAUNCH(poly::divlin_expand_kernel, ngroups, C[j], (const Fr*)V[j], (const Fr*)C[j + 1], (u64)L[j + 1], arg(M[j + 1]));
  }
  KLAUNCH(poly::divlin_final_kernel, L[1], q, p, (const Fr*)C[0], (u64)len, arg(z));
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ---- multi-GPU: MSM sharding by bucket range -------------------------------------------
// Every rank runs the whole prover (AHP rounds replicated: all ranks hold every coefficient vector and the whole window
// table anyway), recodes every scalar, but sorts, accumulates and reduces only the digits that fall into ITS range of
// the 2^(c-1) buckets; its result is a partial sum.  The caller-supplied all_gather (torch.distributed over RCCL/xGMI)
// exchanges the 144-byte partial points and every rank adds them, so all ranks see identical commitments.  Unlike a
// split of the points, this shrinks the sort, the accumulation AND the bucket reduction by the number of ranks and keeps
// the window width of the one-GPU table.
// ag_dev: all-gather of DEVICE buffers (mh_marlin_set_allgather_dev; the native RCCL transport registers ncclAllGather); without it the
// round polynomials travel through the all-to-all with the same chunk for every peer
struct Shard {
  int rank = 0, world = 1; mh_allgather_fn cb = nullptr; void* user = nullptr; mh_alltoall_fn a2a = nullptr; void* a2a_user = nullptr;
  bool a2a_ordered = false; mh_allgather_dev_fn ag_dev = nullptr; void* ag_dev_user = nullptr; bool native = false;
  double host_ms = 0; uint64_t calls = 0;              // wall time the host spent inside exchanges (mh_marlin_exchange_stats)
};
// ---- a rank that fails locally fails the JOB instead of hanging it ---------------------------------------
// Inside a sharded mh_marlin_prove a rank that returned at its first local error would leave its peers blocked in the next
// collective.  Instead it is POISONED: the remaining local steps of the proof are skipped (PTRY), every collective up to the next
// all-gather of partial points is still entered with a payload of the right size (the buffers it would have used, or two of the
// key's scratch vectors when those are what could not be allocated), and that all-gather carries an error word: every rank sees
// it, marks the job failed and returns non-zero from the same commit round -- the transport stays in step and the next proof
// can run.  One GPU (world = 1): nothing changes, the first error returns at once.
struct JobStatus {
  int poison = MH_OK; std::string msg;       // this rank's first local failure
  bool failed = false;                       // some rank's error word came back from an all-gather, or a collective itself failed: every rank is returning
  void* buf[2] = {nullptr, nullptr}; size_t buf_bytes = 0;   // stand-in exchange buffers of a poisoned rank
};
// ---- the in-process transport: several contexts of ONE process as the ranks of a sharded prover (mh_group_*) ------------------
// The reference proves in one process (/root/reference src/lib.rs:151-155, rayon threads: src/ahp/mod.rs:9-10).  With one context
// per GPU and one host thread per context the same shape reaches N GPUs without RCCL and without a launcher: the three exchanges
// of a sharded proof become
//   * all-gather of HOST payloads (partial points, block values): every rank writes its slot of a shared buffer between two
//     rendezvous of the rank threads;
//   * all-to-all / all-gather of DEVICE buffers: every rank publishes its send pointer and an event recorded on its stream, then
//     PULLS its chunks from the peers' buffers with hipMemcpyPeerAsync on its own stream behind those events (xGMI peer-to-peer
//     between GPUs, a plain device copy between contexts that share one) and publishes a second event that the peers' streams
//     wait for before they may overwrite what was pulled -- stream-ordered on every GPU, the host threads only rendezvous.
// A rank that does not arrive within MH_GROUP_TIMEOUT_S (default 300) makes the collective fail on the ranks that wait.
struct LocalGroup {
  int world = 0;
  std::vector<Context*> ctx;                    // the group owns its contexts (mh_group_create / mh_group_destroy)
  std::mutex mu; std::condition_variable cv; int arrived = 0; uint64_t gen = 0; bool broken = false;
  std::vector<uint8_t> host;                    // all-gather of host payloads: world slots of `slot` bytes
  size_t slot = 0;
  std::vector<const void*> send; std::vector<hipEvent_t> ready, done;   // per rank: published send buffer, its two events
  double timeout_s = 300;
  // all ranks arrive, the LAST one runs `last` (under the lock) before anyone leaves
  int rendezvous(const std::function<void()>& last = nullptr) {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return MH_EHIP;
    const uint64_t g = gen;
    if (++arrived == world) { if (last) last(); arrived = 0; gen++; cv.notify_all(); return MH_OK; }
    if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != g || broken; }) || broken) { broken = true; cv.notify_all(); return MH_EHIP; }
    return MH_OK;
  }
};
struct LocalMember { std::shared_ptr<LocalGroup> g; int rank; };

// Everything the prover keeps PER CONTEXT (context.h: Context::ext): prover keys, the shard configuration, the status of the sharded
// proof in flight, the native transport's communicator.  Rounds 1-5 kept these as process globals, which made "one process = one
// GPU = one proof at a time" a property of the library; now two contexts prove side by side (tests/test_gpu_contexts.py) and one
// process can drive several GPUs (mh_marlin_set_local_group).
struct ProverState : mh::ContextExt {
  std::map<uint64_t, std::unique_ptr<ProverKey>> pks;
  Shard shard;
  JobStatus job;
  rcclnative::State rccl;
  LocalMember member{nullptr, 0};                                      // mh_group_create: the in-process transport
  // a context is destroyed while the HIP runtime is alive (mh_shutdown / mh_ctx_destroy); the default context never is
  ~ProverState() override { for (auto& kv : pks) kv.second->free_all(); }
};
inline ProverState& pstate(Context& c) {
  if (!c.ext) c.ext.reset(new ProverState());
  return *static_cast<ProverState*>(c.ext.get());
}
inline ProverState& pstate() { return pstate(ctx()); }
// (the names the code below was written with; every use is under the current context's lock)
#define g_shard (pstate().shard)
#define g_job (pstate().job)
#define g_pks (pstate().pks)
// one exchange: HIP events on the library's stream (family PF_EXCHANGE) and the host's wall clock around it
struct ExchangeScope {
  ProfScope ps; std::chrono::steady_clock::time_point t0;
  explicit ExchangeScope(Context& c) : ps(c, PF_EXCHANGE), t0(std::chrono::steady_clock::now()) {}
  ~ExchangeScope() { g_shard.host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); g_shard.calls++; }
};


inline int group_fail(const char* what) { return fail(MH_EHIP, std::string("in-process transport: ") + what + " (a rank did not arrive, or the group was torn down)"); }
// mh_allgather_fn
int local_allgather_host(const void* send, size_t bytes, void* recv, void* user) {
  LocalMember& m = *static_cast<LocalMember*>(user);
  LocalGroup& g = *m.g;
  if (g.rendezvous([&] { if (g.slot != bytes || g.host.size() != bytes * g.world) { g.slot = bytes; g.host.assign(bytes * g.world, 0); } }) != MH_OK) return group_fail("all-gather");
  if (g.slot != bytes) return fail(MH_EINVAL, "in-process transport: the ranks entered an all-gather with different payload sizes");
  memcpy(g.host.data() + (size_t)m.rank * bytes, send, bytes);
  if (g.rendezvous() != MH_OK) return group_fail("all-gather");
  memcpy(recv, g.host.data(), bytes * g.world);
  if (g.rendezvous() != MH_OK) return group_fail("all-gather");           // nobody resizes or rewrites the slots before everyone has read them
  return MH_OK;
}
// chunk(p) = what this rank takes from peer p's published buffer, and where it puts it
static int local_pull(LocalMember& m, const void* d_send, void* d_recv, size_t take, bool a2a) {
  Context& c = ctx();
  LocalGroup& g = *m.g;
  const int r = m.rank, W = g.world;
  MH_HIP(hipEventRecord(g.ready[r], c.stream));               // my send buffer is final once the stream gets here
  g.send[r] = d_send;
  if (g.rendezvous() != MH_OK) return group_fail("device exchange");
  for (int k = 0; k < W; k++) {
    const int p = (r + k) % W;                                  // staggered: at any moment every link carries one pull
    if (p != r) MH_HIP(hipStreamWaitEvent(c.stream, g.ready[p], 0));
    const char* src = (const char*)g.send[p] + (a2a ? (size_t)r * take : 0);
    char* dst = (char*)d_recv + (size_t)p * take;
    if (g.ctx[p]->device == c.device) MH_HIP(hipMemcpyAsync(dst, src, take, hipMemcpyDeviceToDevice, c.stream));
    else MH_HIP(hipMemcpyPeerAsync(dst, c.device, src, g.ctx[p]->device, take, c.stream));
  }
  MH_HIP(hipEventRecord(g.done[r], c.stream));                // I have pulled everything I need
  if (g.rendezvous() != MH_OK) return group_fail("device exchange");
  for (int p = 0; p < W; p++) if (p != r) MH_HIP(hipStreamWaitEvent(c.stream, g.done[p], 0));   // my send buffer may be rewritten only behind the peers' pulls
  return MH_OK;
}
int local_alltoall_dev(const void* d_send, size_t bytes_per_peer, void* d_recv, void* user) { return local_pull(*static_cast<LocalMember*>(user), d_send, d_recv, bytes_per_peer, true); }
int local_allgather_dev(const void* d_send, size_t bytes, void* d_recv, void* user) { return local_pull(*static_cast<LocalMember*>(user), d_send, d_recv, bytes, false); }
void local_group_leave(Context& c) {
  if (!c.ext) return;
  ProverState& ps = pstate(c);
  if (!ps.member.g) return;
  { std::lock_guard<std::mutex> lk(ps.member.g->mu); ps.member.g->broken = true; ps.member.g->cv.notify_all(); }
  if (ps.shard.user == &ps.member) ps.shard = Shard();
  ps.member.g.reset();
}
inline void poison(int rc) { if (g_job.poison == MH_OK) { g_job.poison = rc; g_job.msg = g_err; } }
inline int job_error() { return g_job.poison != MH_OK ? fail(g_job.poison, g_job.msg) : MH_EHIP; }
// local step: skipped once poisoned; its failure poisons (sharded) or returns (one GPU)
#define PTRY(expr)                                                            \
  do {                                                                        \
    if (g_job.poison == MH_OK) {                                              \
      int _r = (expr);                                                        \
      if (_r != MH_OK) { if (g_shard.world <= 1) return _r; poison(_r); }     \
    }                                                                         \
  } while (0)
// step with a collective inside: always entered; returns when the job failed on every rank, poisons on a failure of this rank only
#define CTRY(expr)                                                            \
  do {                                                                        \
    int _r = (expr);                                                          \
    if (_r != MH_OK) {                                                        \
      if (g_shard.world <= 1) return _r;                                      \
      if (g_job.failed) return g_job.poison != MH_OK ? fail(g_job.poison, g_job.msg) : _r; \
      poison(_r);                                                             \
    }                                                                         \
  } while (0)
#define PHIP(call) PTRY([&]() -> int { MH_HIP(call); return MH_OK; }())

HG1 jac_from(const uint64_t* xyz);
int ntt_dist_device(Context& c, const void* d_in, uint64_t in_len, void* d_out, uint32_t log_n, int inverse);
struct MsmJob { const char* bases; const Fr* scalars; uint64_t n; };
// all jobs in one launch sequence (msm_batch_device) and ONE all_gather for all partial points.
// Payload per rank: nj Jacobian points | one error word | nj flag words (1 = this rank's point is its SHARE of the sum,
// 0 = it is the WHOLE sum).  Which of the two a rank produces is decided rank by rank -- a short vector or a table with
// fewer partitions than ranks is computed in full everywhere, and the skew fallback (largest bucket of the rank's OWN
// partitions) can strike on one rank only: the hot digit lives in exactly one partition.  So the combination rule is
// per job: if any rank reports a whole sum, the lowest such rank's point IS the result (the shares of the others are
// discarded); otherwise the result is the sum of all shares.
int sharded_msm_batch(Context& c, const std::vector<MsmJob>& jobs, std::vector<HG1>& out, int is_mont = 1, bool already_shares = false) {
  const int nj = (int)jobs.size();
  out.assign(nj, HG1::identity());
  if (nj == 0) return MH_OK;
  std::vector<const void*> b(nj), sc(nj); std::vector<size_t> ns(nj);
  for (int j = 0; j < nj; j++) { b[j] = jobs[j].bases; sc[j] = jobs[j].scalars; ns[j] = jobs[j].n; }
  std::vector<uint64_t> part((size_t)XYZ_L * nj);
  const bool sharded = g_shard.world > 1;
  const int sh[2] = {g_shard.rank, g_shard.world};
  std::vector<uint8_t> partial(nj, 0);
  // already_shares: the jobs ARE this rank's part of the sums (blocks of the scalar vectors against the matching base
  // ranges: point sharding) -- computed in full here and flagged as shares for the combination below
  // a poisoned rank (see JobStatus) skips its launches and reports through the error word
  int rc = sharded && g_job.poison != MH_OK ? g_job.poison
                                            : msm_batch_device(c, nj, b.data(), sc.data(), ns.data(), is_mont, part.data(),
                                                               sharded && !already_shares ? sh : nullptr, sharded ? partial.data() : nullptr);
  if (sharded && rc != MH_OK) poison(rc);
  if (sharded && already_shares) std::fill(partial.begin(), partial.end(), (uint8_t)1);
  if (!sharded) { MH_TRY(rc); for (int j = 0; j < nj; j++) out[j] = jac_from(part.data() + XYZ_L * j); return MH_OK; }
  if (!g_shard.cb) return fail(MH_EINVAL, "sharded prove: no all_gather callback registered");
  // a rank whose launch failed still enters the collective (with the error word set) so that the other ranks do not
  // block in it; every rank then fails together
  const size_t npts = part.size(), stride = npts + 1 + (size_t)nj;
  std::vector<uint64_t> send(stride, 0), all(stride * g_shard.world);
  if (rc == MH_OK) {
    memcpy(send.data(), part.data(), npts * 8);
    for (int j = 0; j < nj; j++) send[npts + 1 + j] = partial[j] ? 1 : 0;
  } else {
    send[npts] = 1;
  }
  {
    ExchangeScope xs(c);
    // a collective that FAILS is not a local failure: the communicator is gone for every rank, so the job is marked failed and CTRY
    // returns at once instead of poisoning this rank and entering the next collective on a dead transport
    if (g_shard.cb(send.data(), send.size() * 8, all.data(), g_shard.user) != 0) { g_job.failed = true; return fail(MH_EHIP, "sharded prove: all_gather failed: " + g_err); }
  }
  for (int g = 0; g < g_shard.world; g++) if (all[(size_t)g * stride + npts] != 0) g_job.failed = true;
  if (rc != MH_OK) return fail(rc, g_job.msg);
  if (g_job.failed) return fail(MH_EHIP, "sharded prove: another rank failed (its error word came with the partial points)");
  for (int j = 0; j < nj; j++) {
    int whole = -1;
    for (int g = 0; g < g_shard.world && whole < 0; g++) if (all[(size_t)g * stride + npts + 1 + j] == 0) whole = g;
    if (whole >= 0) { out[j] = jac_from(all.data() + (size_t)whole * stride + XYZ_L * j); continue; }
    for (int g = 0; g < g_shard.world; g++) out[j] = out[j].add(jac_from(all.data() + (size_t)g * stride + XYZ_L * j));
  }
  return MH_OK;
}

// affine normalisation of an MSM result
HG1 jac_from(const uint64_t* xyz) { HG1 p; memcpy(p.X.v, xyz, FQ_B); memcpy(p.Y.v, xyz + FQ_L, FQ_B); memcpy(p.Z.v, xyz + 2 * FQ_L, FQ_B); return p; }

// sum_i scalars[i] * bases[i] for the 3-coefficient hiding polynomials (host): Straus' interleaving -- one shared
// doubling chain, a table of the 2^k - 1 subset sums of the k <= 3 bases
HG1 small_msm(const HG1Affine* bases, const std::vector<HFr>& scalars) {
  const size_t k = scalars.size();
  if (k == 0) return HG1::identity();
  if (k > 3) {
    HG1 acc = HG1::identity();
    for (size_t i = 0; i < k; i++) {
      if (scalars[i].is_zero()) continue;
      uint64_t e[4]; scalars[i].to_canonical(e);
      acc = acc.add(HG1::from_affine(bases[i]).mul(e, 4));
    }
    return acc;
  }
  uint64_t e[3][4];
  for (size_t i = 0; i < k; i++) scalars[i].to_canonical(e[i]);
  HG1 table[8];
  table[0] = HG1::identity();
  for (unsigned m = 1; m < (1u << k); m++) {
    unsigned low = m & (~m + 1);                    // lowest set bit
    int idx = low == 1 ? 0 : (low == 2 ? 1 : 2);
    table[m] = table[m ^ low].add(HG1::from_affine(bases[idx]));
  }
  HG1 acc = HG1::identity();
  for (int limb = 3; limb >= 0; limb--)
    for (int b = 63; b >= 0; b--) {
      acc = acc.dbl();
      unsigned m = 0;
      for (size_t i = 0; i < k; i++) m |= (unsigned)((e[i][limb] >> b) & 1) << i;
      if (m) acc = acc.add(table[m]);
    }
  return acc;
}

HFr host_eval(const std::vector<HFr>& p, const HFr& z) {
  HFr acc = HFr::zero();
  for (size_t i = p.size(); i-- > 0;) acc = acc * z + p[i];
  return acc;
}
std::vector<HFr> host_div_linear(const std::vector<HFr>& p, const HFr& z) {
  std::vector<HFr> q;
  if (p.size() <= 1) return q;
  q.resize(p.size() - 1);
  HFr run = HFr::zero();
  for (size_t i = p.size() - 1; i >= 1; i--) { run = p[i] + z * run; q[i - 1] = run; }
  return q;
}
void host_axpy(std::vector<HFr>& acc, const HFr& f, const std::vector<HFr>& p) {
  if (acc.size() < p.size()) acc.resize(p.size(), HFr::zero());
  for (size_t i = 0; i < p.size(); i++) acc[i] = acc[i] + f * p[i];
}
bool host_is_zero(const std::vector<HFr>& p) { for (auto& x : p) if (!x.is_zero()) return false; return true; }

// The prover's `zk_rng: &mut R` (/root/reference src/lib.rs:151-155).  Every draw `Marlin::prove` makes from it is one
// `F::rand`, so the generator is either the reference's own ChaChaRng, replayed here and on the
// device, or -- for a caller with any other `RngCore` -- the list of field elements that caller drew, in consumption order
// (mh_marlin_prove_draws): r_w, r_za, r_zb, the 3|H| mask coefficients, then 3 per hiding commitment.
struct ZkSource {
  fsh::ChaChaRng chacha;
  const uint64_t* draws = nullptr;        // host memory, Montgomery limbs; nullptr = use chacha
  size_t n = 0, pos = 0;
  bool bad = false;
  HFr fr() {
    if (!draws) return fsh::fr_rand(chacha);
    if (pos >= n) { bad = true; return HFr::zero(); }
    HFr x; memcpy(x.v, draws + 4 * pos, 32); pos++;
    if (HFr::geq_mod(x.v)) bad = true;
    return x;
  }
};

// DensePolynomial::rand(needed - 1, zk_rng) on the device (rng.cuh): out[0..needed) <- the next `needed`
// accepted Fp256::rand draws of the ChaCha stream; advances the host generator past the consumed words.
int device_poly_rand(Context& c, ProverKey& pk, ZkSource& src, Fr* out, uint64_t needed, Fr* cand, u32* flag) {
  if (src.draws) {                        // caller-supplied draws: the next `needed` elements, range-checked on the host
    if (src.pos + needed > src.n) { src.bad = true; return fail(MH_EINVAL, "mh_marlin_prove_draws: too few zk draws"); }
    const uint64_t* p = src.draws + 4 * src.pos;
    for (uint64_t i = 0; i < needed; i++) if (HFr::geq_mod(p + 4 * i)) return fail(MH_EINVAL, "mh_marlin_prove_draws: draw out of range");
    MH_HIP(hipMemcpyAsync(out, p, needed * 32, hipMemcpyHostToDevice, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));
    src.pos += needed;
    return MH_OK;
  }
  fsh::ChaChaRng& zk = src.chacha;
  if (zk.word_pos() % 8) return fail(MH_EINVAL, "zk_rng is not aligned to a field-element draw");
  uint64_t c0 = zk.word_pos() / 8, produced = 0;
  rng::Key key; memcpy(key.k, zk.key, 32);
  u32* sums = (u32*)pk.small.p;
  u32* d_total = (u32*)pk.scal.p;
  u64* d_last = (u64*)((char*)pk.scal.p + 64);
  const uint64_t cap = pk.S[0].bytes / 32;
  while (produced < needed) {
    uint64_t want = needed - produced;
    uint64_t n = want + want / 8 + 4096;
    if (n > cap) n = cap;
    uint64_t nblk = (n + 1023) / 1024;
    ProfScope ps(c, PF_GLUE);
    hipLaunchKernelGGL(rng::candidates_kernel, dim3(poly::grid_for(n / 2 + 2)), dim3(256), 0, c.stream, cand, flag, key, zk.rounds, (u64)c0, (u64)n);
    hipLaunchKernelGGL(rng::scan_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, c.stream, (const u32*)flag, sums, (u64)n);
    hipLaunchKernelGGL(rng::scan_sums_kernel, dim3(1), dim3(1024), 0, c.stream, sums, (u64)nblk, d_total);
    hipLaunchKernelGGL(rng::scan_scatter_kernel, dim3((unsigned)nblk), dim3(256), 0, c.stream, out, (const Fr*)cand, (const u32*)flag,
                       (const u32*)sums, (u64)n, (u64)needed, (u64)produced, d_last);
    MH_HIP(hipGetLastError());
    uint32_t total = 0; uint64_t last = 0;
    MH_HIP(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, c.stream));
    MH_HIP(hipMemcpyAsync(&last, d_last, 8, hipMemcpyDeviceToHost, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));
    if (produced + total >= needed) { c0 += last + 1; produced = needed; }
    else { produced += total; c0 += n; }
  }
  zk.seek_words(c0 * 8);
  return MH_OK;
}

struct Trace {
  bool on; std::chrono::steady_clock::time_point t0, last; Context& c;
  explicit Trace(Context& c_) : c(c_) { on = getenv("MH_TRACE") != nullptr; t0 = last = std::chrono::steady_clock::now(); }
  void mark(const char* label) {
    if (!on) return;
    (void)hipStreamSynchronize(c.stream);
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mh_trace] %-52s %9.3f ms (t=%9.3f)\n", label, std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
};

// transcript / proof bytes of a commitment: marlin_pc::Commitment (comm, presence byte, shifted or identity) or, for
// SonicKZG10, the bare kzg10::Commitment
inline void put_comm(std::vector<uint8_t>& out, const fsh::Commitment& cm, int pc) {
  if (pc == 1) fsh::put_g1(out, cm.comm); else fsh::put_commitment(out, cm);
}

// MarlinKZG10::commit for a list of labeled polynomials (ark-poly-commit marlin_pc / kzg10 [SURVEY B-3, B-4]).
// Per polynomial, in order: KZG10::commit on powers_of_g (drawing a fresh 3-coefficient blinding polynomial from the
// rng when hiding), then, if degree-bounded, a second KZG10::commit on the shifted powers (fresh draws again).
// The rng order equals the reference's sequential loop because the MSMs consume no randomness; all the MSMs of the
// call run as ONE batch.
struct CommitReq { const Fr* poly; uint64_t len; bool has_bound; uint64_t bound; bool hiding; };
int marlin_commit(Context& c, ProverKey& pk, const std::vector<CommitReq>& reqs, ZkSource* rng, std::vector<fsh::Commitment>& comms,
                  std::vector<PolyRand>& rands) {
  auto it = c.bases.find(pk.srs_g);
  if (it == c.bases.end()) return fail(MH_EINVAL, "prover key refers to a freed SRS handle");
  const char* pts = (const char*)it->second.d_points;
  std::vector<MsmJob> jobs;
  comms.assign(reqs.size(), fsh::Commitment());
  rands.assign(reqs.size(), PolyRand());
  if (pk.pc == 1) {
    // SonicKZG10::commit [SURVEY B-5]: one KZG10::commit per polynomial; a degree-bounded one goes against the
    // shifted powers powers_of_g[max_degree - d ..] and its blinding against powers_of_gamma_g[max_degree - d + i]
    for (size_t i = 0; i < reqs.size(); i++) {
      const CommitReq& q = reqs[i];
      uint64_t off = q.has_bound ? pk.srs_max_degree - q.bound : 0;
      if (q.has_bound && q.len > q.bound + 1) return fail(MH_EINVAL, "polynomial exceeds its degree bound");
      if (off + q.len > it->second.n) return fail(MH_EINVAL, "polynomial degree exceeds the SRS");
      jobs.push_back({pts + off * PT_B, q.poly, q.len});
      if (q.hiding) for (int k = 0; k < 3; k++) rands[i].rand.blind.push_back(rng->fr());
    }
    // the 3-coefficient hiding MSMs run on host threads while the device works on the batch
    std::vector<std::future<HG1>> hid(reqs.size());
    for (size_t i = 0; i < reqs.size(); i++)
      if (reqs[i].hiding) {
        const HG1Affine* gp = !reqs[i].has_bound ? pk.gamma_g : (reqs[i].bound == pk.H - 2 ? pk.gamma_g_h : pk.gamma_g_k);
        { const std::vector<HFr>* bl = &rands[i].rand.blind; hid[i] = host_pool().submit([gp, bl] { return small_msm(gp, *bl); }); }
      }
    WaitAll wait_hid; wait_hid.add(hid);
    std::vector<HG1> res;
    MH_TRY(sharded_msm_batch(c, jobs, res));
    std::vector<HG1> fin(reqs.size());
    for (size_t i = 0; i < reqs.size(); i++) {
      fin[i] = res[i];
      if (reqs[i].hiding) fin[i] = fin[i].add(hid[i].get());
    }
    std::vector<HG1Affine> aff(fin.size());
    mh::host_tick("commit: hiding parts added");
    hostff::batch_to_affine(fin.data(), fin.size(), aff.data());        // one inversion for the round's commitments
    mh::host_tick("commit: to affine");
    for (size_t i = 0; i < reqs.size(); i++) { comms[i].comm = aff[i]; comms[i].has_shifted = false; }
    return MH_OK;
  }
  for (size_t i = 0; i < reqs.size(); i++) {
    const CommitReq& q = reqs[i];
    if (q.len > it->second.n) return fail(MH_EINVAL, "polynomial degree exceeds the SRS");
    jobs.push_back({pts, q.poly, q.len});
    if (q.hiding) for (int k = 0; k < 3; k++) rands[i].rand.blind.push_back(rng->fr());   // P::rand(hiding_bound + 1)
    comms[i].has_shifted = q.has_bound; rands[i].has_shifted = q.has_bound;
    if (q.has_bound) {
      if (q.len > q.bound + 1) return fail(MH_EINVAL, "polynomial exceeds its degree bound");
      uint64_t off = pk.srs_max_degree - q.bound;
      if (off + q.len > it->second.n) return fail(MH_EINVAL, "shifted powers exceed the SRS");
      jobs.push_back({pts + off * PT_B, q.poly, q.len});
      if (q.hiding) for (int k = 0; k < 3; k++) rands[i].shifted.blind.push_back(rng->fr());
    }
  }
  // the 3-coefficient hiding MSMs run on host threads while the device works on the batch
  std::vector<std::future<HG1>> hid(reqs.size()), hid_sh(reqs.size());
  for (size_t i = 0; i < reqs.size(); i++)
    if (reqs[i].hiding) {
      { const HG1Affine* gp = pk.gamma_g; const std::vector<HFr>* bl = &rands[i].rand.blind; hid[i] = host_pool().submit([gp, bl] { return small_msm(gp, *bl); }); }
      if (reqs[i].has_bound) { const HG1Affine* gp = pk.gamma_g; const std::vector<HFr>* bl = &rands[i].shifted.blind; hid_sh[i] = host_pool().submit([gp, bl] { return small_msm(gp, *bl); }); }
    }
  WaitAll wait_hid; wait_hid.add(hid); wait_hid.add(hid_sh);
  std::vector<HG1> res;
  MH_TRY(sharded_msm_batch(c, jobs, res));
  std::vector<HG1> fin;
  size_t j = 0;
  for (size_t i = 0; i < reqs.size(); i++) {
    HG1 cm = res[j++];
    if (reqs[i].hiding) cm = cm.add(hid[i].get());
    fin.push_back(cm);
    if (reqs[i].has_bound) {
      HG1 sh = res[j++];
      if (reqs[i].hiding) sh = sh.add(hid_sh[i].get());
      fin.push_back(sh);
    }
  }
  std::vector<HG1Affine> aff(fin.size());
  hostff::batch_to_affine(fin.data(), fin.size(), aff.data());          // one inversion for the round's commitments
  j = 0;
  for (size_t i = 0; i < reqs.size(); i++) {
    comms[i].comm = aff[j++];
    if (reqs[i].has_bound) comms[i].shifted = aff[j++];
  }
  return MH_OK;
}

uint64_t reindex_by_subdomain(uint64_t H, uint64_t X, uint64_t index) {
  uint64_t period = H / X;
  if (index < X) return index * period;
  uint64_t i = index - X, x = period - 1;
  return i + (i / x) + 1;
}

}  // namespace

namespace mh {
int ensure_twiddles_public(Context& c, uint32_t log_n);
}

// ===========================================================================================
// C ABI
// ===========================================================================================
#define LOCKED_CTX()                                                 \
  Context& c = ctx();                                                \
  CtxLock _lk(c);                                                    \
  if (!c.inited) return fail(MH_ENOINIT, "mh_init has not been called")

extern "C" {

// called by mh_shutdown: release every prover key
int mh_marlin_release_all(void) {
  Context& c = ctx();
  if (!c.ext) return MH_OK;
  for (auto& kv : g_pks) kv.second->free_all();
  g_pks.clear();
  if (g_shard.native) g_shard = Shard();
  (void)rcclnative::destroy(pstate(c).rccl);
  local_group_leave(c);
  g_job = JobStatus();
  return MH_OK;
}

int mh_marlin_set_shard(int rank, int world, mh_allgather_fn allgather, void* user) {
  if (world < 1 || rank < 0 || rank >= world) return fail(MH_EINVAL, "mh_marlin_set_shard: bad rank/world");
  if (world > 1 && !allgather) return fail(MH_EINVAL, "mh_marlin_set_shard: all_gather callback required for world > 1");
  Context& c = ctx();
  CtxLock lk(c);     // g_shard is read by a running mh_marlin_prove under this lock
  // a callback transport replaces the native one (mh_marlin_set_rccl) as a whole: its all-to-all and device all-gather go with
  // it and its communicator is destroyed -- the sliced rounds must never mix a callback all-gather with native all-to-alls
  if (g_shard.native) {
    if (c.inited && c.stream) (void)hipStreamSynchronize(c.stream);
    g_shard.a2a = nullptr; g_shard.a2a_user = nullptr; g_shard.a2a_ordered = false;
    (void)rcclnative::destroy(pstate(c).rccl);
  }
  g_shard.rank = rank; g_shard.world = world; g_shard.cb = allgather; g_shard.user = user;
  g_shard.ag_dev = nullptr; g_shard.ag_dev_user = nullptr; g_shard.native = false;
  return MH_OK;                                 // the window table does not depend on the number of ranks (bucket-range sharding)
}

// ---- native transport: RCCL called by the library on its own stream (rccl_native.h) --------------------------------------
int mh_rccl_unique_id(uint8_t* id128_out) {
  if (!id128_out) return fail(MH_EINVAL, "mh_rccl_unique_id: null output");
  return rcclnative::unique_id(id128_out);
}
// Collective: every rank of the node calls it with the same id.  Registers the all-gather of partial points, the all-to-all of
// the distributed transforms and the device all-gather of round polynomials in one step (what mh_marlin_set_shard +
// mh_marlin_set_alltoall + mh_marlin_set_alltoall_mode(1) do for a callback transport).
int mh_marlin_set_rccl(int rank, int world, const uint8_t* id128) {
  if (world < 1 || rank < 0 || rank >= world || !id128) return fail(MH_EINVAL, "mh_marlin_set_rccl: bad rank / world / id");
  Context& c = ctx();
  CtxLock lk(c);
  if (!c.inited) return fail(MH_ENOINIT, "mh_init has not been called");
  MH_HIP(hipStreamSynchronize(c.stream));
  MH_TRY(rcclnative::init(c, pstate(c).rccl, rank, world, id128));
  void* st = &pstate(c).rccl;
  g_shard.rank = rank; g_shard.world = world; g_shard.cb = rcclnative::allgather_host; g_shard.user = st;
  g_shard.a2a = rcclnative::alltoall_dev; g_shard.a2a_user = st; g_shard.a2a_ordered = true;
  g_shard.ag_dev = rcclnative::allgather_dev; g_shard.ag_dev_user = st; g_shard.native = true;
  return MH_OK;
}
// sliced != 0 (default after mh_marlin_set_rccl): rounds 2 and 3 and the openings run on slices when the geometry allows it;
// 0 keeps the AHP rounds replicated (only the MSMs are sharded) -- what a callback transport gets without an all-to-all
int mh_marlin_rccl_sliced(int sliced) {
  Context& c = ctx();
  CtxLock lk(c);
  if (!g_shard.native) return fail(MH_EINVAL, "mh_marlin_rccl_sliced: the native transport is not active");
  g_shard.a2a = sliced ? rcclnative::alltoall_dev : nullptr; g_shard.a2a_user = &pstate(c).rccl;
  g_shard.a2a_ordered = sliced != 0;
  g_shard.ag_dev = sliced ? rcclnative::allgather_dev : nullptr; g_shard.ag_dev_user = &pstate(c).rccl;
  return MH_OK;
}
int mh_marlin_rccl_destroy(void) {
  Context& c = ctx();
  CtxLock lk(c);
  if (c.inited && c.stream) (void)hipStreamSynchronize(c.stream);
  if (g_shard.native) { g_shard = Shard(); }
  return rcclnative::destroy(pstate(c).rccl);
}
// info4: all-gathers of host payloads, all-to-alls, device all-gathers, bytes this rank sent to peers -- since mh_marlin_set_rccl;
// lib_path (may be NULL): which librccl the symbols came from
int mh_marlin_rccl_info(uint64_t* info4, char* lib_path, size_t cap) {
  const rcclnative::State& s = pstate().rccl;
  if (info4) { info4[0] = s.n_allgather_host; info4[1] = s.n_alltoall; info4[2] = s.n_allgather_dev; info4[3] = s.bytes_moved; }
  if (lib_path && cap) snprintf(lib_path, cap, "%s", rcclnative::api().path.c_str());
  return g_shard.native ? 1 : 0;
}
// exchanges since the last reset, whatever the transport: count and the host's wall-clock milliseconds inside them
int mh_marlin_exchange_stats(uint64_t* calls_out, double* host_ms_out, int reset) {
  Context& c = ctx();
  CtxLock lk(c);
  if (calls_out) *calls_out = g_shard.calls;
  if (host_ms_out) *host_ms_out = g_shard.host_ms;
  if (reset) { g_shard.calls = 0; g_shard.host_ms = 0; }
  return MH_OK;
}

// ---- one process, several GPUs: a group of contexts joined by the in-process transport (LocalGroup above) ----------------------
struct mh_group { std::shared_ptr<LocalGroup> g; std::vector<mh_ctx_t> handles; };
int mh_group_create(const int* device_ids, int world, mh_group_t* group_out) {
  if (!device_ids || !group_out || world < 1 || world > 64) return fail(MH_EINVAL, "mh_group_create: bad argument (1 <= world <= 64)");
  *group_out = nullptr;
  std::unique_ptr<mh_group> grp(new mh_group());
  grp->g = std::make_shared<LocalGroup>();
  LocalGroup& g = *grp->g;
  g.world = world;
  { const char* e = getenv("MH_GROUP_TIMEOUT_S"); if (e && atof(e) > 0) g.timeout_s = atof(e); }
  g.send.assign(world, nullptr); g.ready.assign(world, nullptr); g.done.assign(world, nullptr);
  int prev = -1;
  (void)hipGetDevice(&prev);
  auto undo = [&](int rc) { for (auto h : grp->handles) (void)mh_ctx_destroy(h); for (auto e : g.ready) if (e) (void)hipEventDestroy(e); for (auto e : g.done) if (e) (void)hipEventDestroy(e);
                            if (prev >= 0) (void)hipSetDevice(prev); return rc; };
  for (int r = 0; r < world; r++) {
    mh_ctx_t h = nullptr;
    const int rc = mh_ctx_create(device_ids[r], &h);
    if (rc != MH_OK) return undo(rc);
    grp->handles.push_back(h);
    g.ctx.push_back(reinterpret_cast<Context*>(h));
    if (hipSetDevice(device_ids[r]) != hipSuccess || hipEventCreateWithFlags(&g.ready[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g.done[r], hipEventDisableTiming) != hipSuccess) return undo(fail(MH_EHIP, "mh_group_create: hipEventCreate failed"));
  }
  // peer-to-peer between the group's GPUs (xGMI): enabled where the hardware allows it; hipMemcpyPeerAsync stages through the host otherwise
  for (int a = 0; a < world; a++)
    for (int b = 0; b < world; b++) {
      const int da = device_ids[a], db = device_ids[b];
      int can = 0;
      if (da == db || hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) continue;
      (void)hipSetDevice(da);
      const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
      if (e != hipSuccess) (void)hipGetLastError();            // already enabled (another group, the caller): fine
    }
  if (prev >= 0) (void)hipSetDevice(prev);
  const bool pow2 = (world & (world - 1)) == 0 && world <= 8;
  for (int r = 0; r < world; r++) {
    Context& c = *g.ctx[r];
    CtxLock lk(c);
    ProverState& ps = pstate(c);
    ps.member = LocalMember{grp->g, r};
    ps.shard = Shard();
    ps.shard.rank = r; ps.shard.world = world; ps.shard.cb = local_allgather_host; ps.shard.user = &ps.member;
    if (pow2 && world > 1) { ps.shard.a2a = local_alltoall_dev; ps.shard.a2a_user = &ps.member; ps.shard.a2a_ordered = true;
                             ps.shard.ag_dev = local_allgather_dev; ps.shard.ag_dev_user = &ps.member; }
  }
  *group_out = grp.release();
  return MH_OK;
}
int mh_group_size(mh_group_t grp) { return grp ? grp->g->world : 0; }
mh_ctx_t mh_group_ctx(mh_group_t grp, int rank) { return grp && rank >= 0 && rank < grp->g->world ? grp->handles[rank] : nullptr; }
// fn(rank, user) on `world` threads, thread r bound to the group's context r; returns the first non-zero result (its message in mh_last_error)
int mh_group_run(mh_group_t grp, mh_group_fn fn, void* user) {
  if (!grp || !fn) return fail(MH_EINVAL, "mh_group_run: null argument");
  const int W = grp->g->world;
  std::vector<int> rc(W, MH_OK); std::vector<std::string> msg(W);
  std::vector<std::thread> th;
  for (int r = 0; r < W; r++)
    th.emplace_back([&, r] {
      rc[r] = mh_ctx_set_current(grp->handles[r]);
      if (rc[r] == MH_OK) rc[r] = fn(r, user);
      if (rc[r] != MH_OK) msg[r] = g_err;
      (void)mh_ctx_set_current(nullptr);
    });
  for (auto& t : th) t.join();
  for (int r = 0; r < W; r++) if (rc[r] != MH_OK) return fail(rc[r], "rank " + std::to_string(r) + ": " + msg[r]);
  return MH_OK;
}
int mh_group_destroy(mh_group_t grp) {
  if (!grp) return MH_OK;
  LocalGroup& g = *grp->g;
  { std::lock_guard<std::mutex> lk(g.mu); g.broken = true; g.cv.notify_all(); }
  int rc = MH_OK;
  for (auto h : grp->handles) { const int r = mh_ctx_destroy(h); if (rc == MH_OK) rc = r; }
  for (auto e : g.ready) if (e) (void)hipEventDestroy(e);
  for (auto e : g.done) if (e) (void)hipEventDestroy(e);
  delete grp;
  return rc;
}

// ---- distributed building blocks (the slice-sharded pipeline) ---------------------------------------------
int mh_marlin_set_alltoall(mh_alltoall_fn alltoall, void* user) {
  Context& c = ctx();
  CtxLock lk(c);
  g_shard.a2a = alltoall; g_shard.a2a_user = user; g_shard.a2a_ordered = false;
  g_shard.ag_dev = nullptr; g_shard.ag_dev_user = nullptr;
  return MH_OK;
}
// optional, after mh_marlin_set_alltoall: a real all-gather of device buffers for the round polynomials of the sliced sections
// (otherwise they go through the all-to-all, every rank sending `world` copies of its chunk); follows the all-to-all's mode
int mh_marlin_set_allgather_dev(mh_allgather_dev_fn allgather_dev, void* user) {
  Context& c = ctx();
  CtxLock lk(c);
  g_shard.ag_dev = allgather_dev; g_shard.ag_dev_user = user;
  return MH_OK;
}
// stream_ordered != 0: the registered all-to-all enqueues its work on the library's stream (mh_set_stream: the caller's stream) or
// orders itself against it on the device, so the library does not drain the stream before calling it.  Reset by mh_marlin_set_alltoall.
int mh_marlin_set_alltoall_mode(int stream_ordered) {
  Context& c = ctx();
  CtxLock lk(c);
  g_shard.a2a_ordered = stream_ordered != 0;
  return MH_OK;
}

}  // extern "C"
namespace {
// the transform itself; forward accepts a SHORT slice (in_len <= n / world local coefficients, the rest reads as zero)
int ntt_dist_device(Context& c, const void* d_in, uint64_t in_len, void* d_out, uint32_t log_n, int inverse) {
  const int G = g_shard.world, rank = g_shard.rank;
  uint32_t lg = 0;
  while ((1 << lg) < G) lg++;
  if ((1 << lg) != G || G > 8) return fail(MH_EINVAL, "mh_ntt_dist_dev: the number of ranks must be a power of two <= 8 (one node)");
  if (!g_shard.a2a) return fail(MH_EINVAL, "mh_ntt_dist_dev: no all_to_all callback registered (mh_marlin_set_alltoall)");
  if (log_n > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "log_n exceeds the two-adicity of Fr");
  if (log_n < 2 * lg) return fail(MH_EINVAL, "mh_ntt_dist_dev: the transform must have at least world^2 points");
  const uint32_t log_m = log_n - lg;
  const uint64_t m = 1ull << log_m, chunk = m >> lg;
  // lrc: this rank's local status.  Once it is not MH_OK the local kernels are skipped but the exchange below is still entered
  // (JobStatus): the peers are already on their way into it.
  int lrc = g_job.poison;
  auto L = [&](int r) { if (lrc == MH_OK && r != MH_OK) { lrc = r; poison(r); } };
  L(c.ntt_dist_buf[0].ensure(m * 32));
  if (lrc == MH_OK) L(c.ntt_dist_buf[1].ensure(m * 32));
  Fr* A = (Fr*)c.ntt_dist_buf[0].ptr; Fr* B = (Fr*)c.ntt_dist_buf[1].ptr;
  if (c.ntt_dist_buf[0].cap < m * 32 || c.ntt_dist_buf[1].cap < m * 32) {       // what failed is these very buffers
    if (g_job.buf_bytes < m * 32) { g_job.failed = true; return fail(MH_ENOMEM, "mh_ntt_dist_dev: no exchange buffer (this rank cannot enter the all-to-all)"); }
    A = (Fr*)g_job.buf[0]; B = (Fr*)g_job.buf[1];
  }
  // w_n, and the tables of (w_n^(+-rank))^k2 = hi[k2 >> 11] * lo[k2 & 2047], the inverse's lo carrying G^-1
  HFr wn = hostff::fr_two_adic_root();
  for (uint32_t i = log_n; i < hostff::FR_TWO_ADICITY_H; i++) wn = wn.sqr();
  if (inverse) wn = wn.inv();
  const uint64_t LO = 1ull << poly::COSET_LO_BITS, nhi = (m + LO - 1) / LO;
  const uint64_t tab_key = ((uint64_t)log_n << 16) | ((uint64_t)(inverse ? 1 : 0) << 15) | ((uint64_t)G << 8) | (uint64_t)rank;
  Scratch& tabs = c.ntt_dist_tabs[tab_key];
  if (!tabs.ptr && lrc == MH_OK) {                  // built once per (size, direction, geometry): ~2300 host multiplications
    const HFr base = wn.pow_u64((uint64_t)rank);
    std::vector<uint64_t> tab(4 * (LO + nhi));
    HFr a = inverse ? HFr::from_u64((uint64_t)G).inv() : HFr::one();
    for (uint64_t j = 0; j < LO; j++) { memcpy(&tab[4 * j], a.v, 32); a = a * base; }
    const HFr bl = base.pow_u64(LO);
    a = HFr::one();
    for (uint64_t j = 0; j < nhi; j++) { memcpy(&tab[4 * (LO + j)], a.v, 32); a = a * bl; }
    L(tabs.ensure(tab.size() * 8));
    if (lrc == MH_OK) {
      MH_HIP(hipMemcpyAsync(tabs.ptr, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c.stream));
      MH_HIP(hipStreamSynchronize(c.stream));        // `tab` is pageable and goes out of scope
    } else tabs.release();
  }
  const Fr* t_lo = (const Fr*)tabs.ptr; const Fr* t_hi = t_lo + LO;
  nttdist::Roots roots;
  {
    HFr wg = wn.pow_u64(m), a = HFr::one();          // w_G = w_n^m (the inverse root when `inverse`)
    for (int e = 0; e < 8; e++) { roots.w[e] = dfr(a); a = a * wg; }
  }
  auto gdft = [&](Fr* out, const Fr* in) -> int {
    ProfScope ps(c, PF_NTT);
    const dim3 grid((unsigned)((chunk + 255) / 256)), block(256);
    switch (lg) {
      case 1: hipLaunchKernelGGL(nttdist::gdft_kernel<1>, grid, block, 0, c.stream, out, in, (u64)chunk, roots); break;
      case 2: hipLaunchKernelGGL(nttdist::gdft_kernel<2>, grid, block, 0, c.stream, out, in, (u64)chunk, roots); break;
      default: hipLaunchKernelGGL(nttdist::gdft_kernel<3>, grid, block, 0, c.stream, out, in, (u64)chunk, roots); break;
    }
    MH_HIP(hipGetLastError());
    return MH_OK;
  };
  auto twist = [&](Fr* buf) -> int {
    ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::twist_kernel, m, buf, (const Fr*)buf, t_hi, t_lo, (u64)m);
    MH_HIP(hipGetLastError());
    return MH_OK;
  };
  // the exchange sees finished buffers and hands back a finished buffer (the transport may use its own stream)
  auto exchange = [&](const Fr* send, Fr* recv) -> int {
    if (!g_shard.a2a_ordered) MH_HIP(hipStreamSynchronize(c.stream));
    ExchangeScope xs(c);
    if (g_shard.a2a(send, (size_t)chunk * 32, recv, g_shard.a2a_user) != 0) { g_job.failed = true; return fail(MH_EHIP, "mh_ntt_dist_dev: all_to_all failed: " + g_err); }
    return MH_OK;
  };
  if (!inverse) {
    if (lrc == MH_OK) L(ntt_device_len(c, d_in, in_len, A, log_m, 0));   // Y_rank = local transform of the cyclic slice (short input: zero-padded)
    if (lrc == MH_OK) L(twist(A));                   // * w_n^(rank k2)
    MH_TRY(exchange(A, B));                          // chunk q = k2 in block q -> rank q (always entered)
    if (lrc == MH_OK) L(gdft((Fr*)d_out, B));        // X[k2 + m k1], k1-major
  } else {
    if (lrc == MH_OK) L(gdft(A, (const Fr*)d_in));   // G Z_j1[k2] for k2 in this rank's block, j1-major
    MH_TRY(exchange(A, B));                          // chunk j1 -> rank j1: B = G w^(j1 k2) Y_j1, k2 in order (always entered)
    if (lrc == MH_OK) L(twist(B));                   // * w_n^(-rank k2) / G
    if (lrc == MH_OK) L(ntt_device(c, B, d_out, log_m, 1));   // local inverse (with m^-1)
  }
  return lrc == MH_OK ? MH_OK : fail(lrc, g_job.msg);
}
}  // namespace
namespace {
// ---- sliced sections of the prover ----------------------------------------------------------------------
// The 4H- and K-sized transforms of rounds 2 and 3 and the pointwise work between them run on SLICES: every rank takes its
// cyclic slice of the (replicated) coefficient vectors, the transforms are distributed (one all-to-all each), the pointwise
// kernels see 1 / G of the domain, and one all-gather per round brings the round's polynomials back to every rank for the
// bucket-sharded commitments.  Proof bytes are unchanged: field arithmetic is exact whoever does it.
inline unsigned grid256(uint64_t n) { return (unsigned)((n + 255) / 256); }

// this rank's cyclic slice of a replicated coefficient vector of `len` elements; returns its length
uint64_t slice_c(Context& c, Fr* dst, const Fr* src, uint64_t len) {
  const uint64_t G = (uint64_t)g_shard.world, r = (uint64_t)g_shard.rank;
  const uint64_t nloc = len > r ? (len - r + G - 1) / G : 0;
  if (nloc && g_job.poison == MH_OK) hipLaunchKernelGGL(nttdist::slice_c_kernel, dim3(grid256(nloc)), dim3(256), 0, c.stream, dst, src, (u64)nloc, (u32)r, (u32)G);
  return nloc;
}
// all ranks' (a | b) chunks, rank-major: the all-to-all with the same chunk for every peer
int allgather2(Context& c, const Fr* a, uint64_t na, const Fr* b, uint64_t nb, const Fr** recv_out) {
  const uint64_t G = (uint64_t)g_shard.world, ch = na + nb;
  const uint64_t copies = g_shard.ag_dev ? 1 : G;        // a real all-gather sends its chunk once
  // a poisoned rank (JobStatus) still enters the collective, with whatever its buffers hold
  int lrc = g_job.poison;
  auto L = [&](int r) { if (lrc == MH_OK && r != MH_OK) { lrc = r; poison(r); } };
  L(c.sl_send.ensure(copies * ch * 32));
  if (lrc == MH_OK) L(c.sl_recv.ensure(G * ch * 32));
  void* snd = c.sl_send.ptr; void* rcv = c.sl_recv.ptr;
  if (c.sl_send.cap < copies * ch * 32 || c.sl_recv.cap < G * ch * 32) {
    if (g_job.buf_bytes < G * ch * 32) { g_job.failed = true; return fail(MH_ENOMEM, "sliced prove: no exchange buffer (this rank cannot enter the all-gather)"); }
    snd = g_job.buf[0]; rcv = g_job.buf[1];
  }
  if (lrc == MH_OK) {
    hipLaunchKernelGGL(nttdist::replicate_kernel, dim3(grid256(ch)), dim3(256), 0, c.stream, (Fr*)snd, a, (u64)na, b, (u64)nb, (u32)copies);
    MH_HIP(hipGetLastError());
  }
  if (!g_shard.a2a_ordered) MH_HIP(hipStreamSynchronize(c.stream));
  {
    ExchangeScope xs(c);
    if (g_shard.ag_dev) {
      if (g_shard.ag_dev(snd, (size_t)ch * 32, rcv, g_shard.ag_dev_user) != 0) { g_job.failed = true; return fail(MH_EHIP, "sliced prove: all_gather failed: " + g_err); }
    } else if (g_shard.a2a(snd, (size_t)ch * 32, rcv, g_shard.a2a_user) != 0) {
      g_job.failed = true;
      return fail(MH_EHIP, "sliced prove: all_to_all failed: " + g_err);
    }
  }
  *recv_out = (const Fr*)rcv;
  return lrc == MH_OK ? MH_OK : fail(lrc, g_job.msg);
}
void unslice_c(Context& c, Fr* full, const Fr* recv, uint64_t len, uint64_t stride, uint64_t off) {
  if (g_job.poison != MH_OK || !recv) return;        // a poisoned rank launches nothing more (its receive buffer may not exist)
  hipLaunchKernelGGL(nttdist::unslice_c_kernel, dim3(grid256(len)), dim3(256), 0, c.stream, full, recv, (u64)len, (u32)g_shard.world, (u64)stride, (u64)off);
}

// per-key, per-geometry data of the sliced sections: M-layout blocks of the index evaluation vectors the third round reads,
// and the twist tables g^(+-(r + G j)) = lo[j & 2047] * hi[j >> 11]
int prepare_sliced(Context& c, ProverKey& pk) {
  const int G = g_shard.world, r = g_shard.rank;
  if (pk.sl_rank == r && pk.sl_world == G) return MH_OK;
  const uint64_t K = pk.K, m = K / (uint64_t)G, chunk = m / (uint64_t)G;
  const DBuf* ev_src[5] = {&pk.ev_row, &pk.ev_col, &pk.ev_val_a, &pk.ev_val_b, &pk.ev_val_c};
  const DBuf* cs_src[6] = {&pk.cs_val_a, &pk.cs_val_b, &pk.cs_val_c, &pk.cs_row, &pk.cs_col, &pk.cs_row_col};
  for (int q = 0; q < 5; q++) {
    MH_TRY(pk.sl_ev[q].alloc(m * 32));
    hipLaunchKernelGGL(nttdist::gather_m_kernel, dim3(grid256(m)), dim3(256), 0, c.stream, pk.sl_ev[q].fr(), (const Fr*)ev_src[q]->p, (u64)m, (u64)chunk, (u32)r);
  }
  for (int q = 0; q < 6; q++) {
    MH_TRY(pk.sl_cs[q].alloc(m * 32));
    hipLaunchKernelGGL(nttdist::gather_m_kernel, dim3(grid256(m)), dim3(256), 0, c.stream, pk.sl_cs[q].fr(), (const Fr*)cs_src[q]->p, (u64)m, (u64)chunk, (u32)r);
  }
  MH_HIP(hipGetLastError());
  const uint64_t LO = 1ull << poly::COSET_LO_BITS, nhi = (m + LO - 1) / LO;
  std::vector<uint64_t> lo(4 * LO), hi(4 * nhi), ilo(4 * LO), ihi(4 * nhi);
  const HFr g = pk.coset_g, ginv = g.inv();
  const HFr gG = g.pow_u64((uint64_t)G), giG = ginv.pow_u64((uint64_t)G);
  HFr a = g.pow_u64((uint64_t)r), b = ginv.pow_u64((uint64_t)r);
  for (uint64_t j = 0; j < LO; j++) { memcpy(&lo[4 * j], a.v, 32); memcpy(&ilo[4 * j], b.v, 32); a = a * gG; b = b * giG; }
  const HFr gL = gG.pow_u64(LO), giL = giG.pow_u64(LO);
  a = HFr::one(); b = HFr::one();
  for (uint64_t j = 0; j < nhi; j++) { memcpy(&hi[4 * j], a.v, 32); memcpy(&ihi[4 * j], b.v, 32); a = a * gL; b = b * giL; }
  MH_TRY(pk.sl_gpow_lo.alloc(LO * 32)); MH_TRY(pk.sl_ginv_lo.alloc(LO * 32)); MH_TRY(pk.sl_gpow_hi.alloc(nhi * 32)); MH_TRY(pk.sl_ginv_hi.alloc(nhi * 32));
  MH_TRY(h2d(c, pk.sl_gpow_lo.p, lo.data(), LO * 32)); MH_TRY(h2d(c, pk.sl_ginv_lo.p, ilo.data(), LO * 32));
  MH_TRY(h2d(c, pk.sl_gpow_hi.p, hi.data(), nhi * 32)); MH_TRY(h2d(c, pk.sl_ginv_hi.p, ihi.data(), nhi * 32));
  MH_HIP(hipStreamSynchronize(c.stream));
  pk.sl_rank = r; pk.sl_world = G;
  return MH_OK;
}
}  // namespace

extern "C" {

// One transform of 2^log_n points over the registered ranks (ntt_dist.cuh).  forward: d_in = this rank's cyclic slice of the
// coefficients (C-layout), d_out = its block of the evaluations (M-layout); inverse: the other way round.  n / world elements
// each; d_in may equal d_out.
int mh_ntt_dist_dev(int field, const void* d_in, void* d_out, uint32_t log_n, int inverse) {
  LOCKED_CTX();
  if (field != hostff::CURVE_ID) return fail(MH_EINVAL, "mh_ntt_dist_dev: unsupported field");
  if (!d_in || !d_out) return fail(MH_EINVAL, "mh_ntt_dist_dev: null pointer");
  if (g_shard.world == 1) return ntt_device(c, d_in, d_out, log_n, inverse);
  g_job = JobStatus();                           // a standalone call is its own job
  return ntt_dist_device(c, d_in, (1ull << log_n) / (uint64_t)g_shard.world, d_out, log_n, inverse);
}

// MSMs of this rank's CYCLIC slices (ntt_dist.cuh's C-layout: scalar i of job j multiplies base first_index[j] + i * stride
// of the handle's set, whose window table serves as is) -- the point-sharded MSM of the slice-sharded pipeline.  combine != 0:
// the partial points of all registered ranks are all-gathered and summed, every rank returns the complete results.
int mh_msm_batch_sliced_dev(uint64_t bases_handle, size_t njobs, const size_t* first_index, size_t stride, const void* const* d_scalars,
                            const size_t* ns, int is_mont, int combine, uint64_t* out_xyz) {
  LOCKED_CTX();
  if (njobs && (!first_index || !d_scalars || !ns || !out_xyz)) return fail(MH_EINVAL, "mh_msm_batch_sliced_dev: null pointer");
  auto it = c.bases.find(bases_handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_msm_batch_sliced_dev: unknown bases handle");
  std::vector<uint64_t> part((size_t)XYZ_L * njobs + 1, 0);
  int rc = msm_batch_strided_device(c, it->second, (int)njobs, first_index, stride, d_scalars, ns, is_mont, part.data());
  if (!combine || g_shard.world == 1) { MH_TRY(rc); memcpy(out_xyz, part.data(), (size_t)XYZ_L * njobs * 8); return MH_OK; }
  if (!g_shard.cb) return fail(MH_EINVAL, "mh_msm_batch_sliced_dev: no all_gather callback registered");
  part[(size_t)XYZ_L * njobs] = rc == MH_OK ? 0 : 1;                 // a failing rank still enters the collective
  std::vector<uint64_t> all(part.size() * g_shard.world);
  {
    ExchangeScope xs(c);
    if (g_shard.cb(part.data(), part.size() * 8, all.data(), g_shard.user) != 0) return fail(MH_EHIP, "mh_msm_batch_sliced_dev: all_gather failed: " + g_err);
  }
  MH_TRY(rc);
  for (int g = 0; g < g_shard.world; g++)
    if (all[(size_t)g * part.size() + XYZ_L * njobs] != 0) return fail(MH_EHIP, "mh_msm_batch_sliced_dev: the MSM of another rank failed");
  for (size_t j = 0; j < njobs; j++) {
    HG1 acc = HG1::identity();
    for (int g = 0; g < g_shard.world; g++) acc = acc.add(jac_from(all.data() + (size_t)g * part.size() + XYZ_L * j));
    uint64_t* o = out_xyz + XYZ_L * j;
    memcpy(o, acc.X.v, FQ_B); memcpy(o + FQ_L, acc.Y.v, FQ_B); memcpy(o + 2 * FQ_L, acc.Z.v, FQ_B);
  }
  return MH_OK;
}

// mh_msm_batch_dev across the ranks registered with mh_marlin_set_shard: every rank passes the SAME jobs (full scalar
// vectors, full base sets), computes its bucket range of each and receives the combined results
int mh_msm_batch_sharded_dev(size_t njobs, const uint64_t* handles, const size_t* base_offsets, const void* const* d_scalars,
                             const size_t* ns, int is_mont, uint64_t* out_xyz) {
  LOCKED_CTX();
  if (njobs && (!handles || !base_offsets || !d_scalars || !ns || !out_xyz)) return fail(MH_EINVAL, "mh_msm_batch_sharded_dev: null pointer");
  std::vector<MsmJob> jobs(njobs);
  for (size_t j = 0; j < njobs; j++) {
    auto it = c.bases.find(handles[j]);
    if (it == c.bases.end()) return fail(MH_EINVAL, "mh_msm_batch_sharded_dev: unknown bases handle");
    if (base_offsets[j] > it->second.n || ns[j] > it->second.n - base_offsets[j])
      return fail(MH_EINVAL, "mh_msm_batch_sharded_dev: base_offset + n exceeds the uploaded base set");
    if (ns[j] && !d_scalars[j]) return fail(MH_EINVAL, "mh_msm_batch_sharded_dev: null scalars");
    jobs[j] = {(const char*)it->second.d_points + base_offsets[j] * PT_B, (const Fr*)d_scalars[j], ns[j]};
  }
  std::vector<HG1> res;
  g_job = JobStatus();                           // a standalone call is its own job
  const int rc = sharded_msm_batch(c, jobs, res, is_mont);
  g_job = JobStatus();
  MH_TRY(rc);
  for (size_t j = 0; j < njobs; j++) {
    uint64_t* o = out_xyz + XYZ_L * j;
    memcpy(o, res[j].X.v, FQ_B); memcpy(o + FQ_L, res[j].Y.v, FQ_B); memcpy(o + 2 * FQ_L, res[j].Z.v, FQ_B);
  }
  return MH_OK;
}

// ---- verifier (host only; verify_host.h, pairing_host.h) -----------------------------------------------------
static bool load_g1(const uint64_t* p, hostff::HG1Affine* out) {
  memcpy(out->x.v, p, FQ_B); memcpy(out->y.v, p + FQ_L, FQ_B); out->inf = false;
  return !HFq::geq_mod(out->x.v) && !HFq::geq_mod(out->y.v) && hostverify::g1_on_curve(*out);
}
static bool load_g2(const uint64_t* p, hostpair::G2Aff* out) {
  memcpy(out->x.a.v, p, FQ_B); memcpy(out->x.b.v, p + FQ_L, FQ_B); memcpy(out->y.a.v, p + 2 * FQ_L, FQ_B); memcpy(out->y.b.v, p + 3 * FQ_L, FQ_B);
  out->inf = false;
  for (const HFq* f : {&out->x.a, &out->x.b, &out->y.a, &out->y.b}) if (HFq::geq_mod(f->v)) return false;
  return hostpair::g2_on_curve(*out);
}
static int marlin_verify_impl(const uint8_t* vk_bytes, size_t vk_len, const mh_verifier_key* vk, int pc, const uint64_t* public_input, size_t n_public,
                              const uint8_t* flat_proof, size_t proof_len, int* ok_out, const fsh::ExternalFs* ext_fs);
int mh_marlin_verify(const uint8_t* vk_bytes, size_t vk_len, const mh_verifier_key* vk, int pc, const uint64_t* public_input, size_t n_public,
                     const uint8_t* flat_proof, size_t proof_len, int* ok_out) {
  return marlin_verify_impl(vk_bytes, vk_len, vk, pc, public_input, n_public, flat_proof, proof_len, ok_out, nullptr);
}
// Marlin<F, PC, FS>::verify with the caller's FS (the counterpart of mh_marlin_prove_fs)
int mh_marlin_verify_fs(const uint8_t* vk_bytes, size_t vk_len, const mh_verifier_key* vk, int pc, const uint64_t* public_input, size_t n_public,
                        const uint8_t* flat_proof, size_t proof_len, const mh_fiat_shamir* fs, int* ok_out) {
  if (!fs || !fs->initialize || !fs->absorb || !fs->next_u64) return fail(MH_EINVAL, "mh_marlin_verify_fs: initialize, absorb and next_u64 must all be set");
  const fsh::ExternalFs ext{fs->user, fs->initialize, fs->absorb, fs->next_u64};
  return marlin_verify_impl(vk_bytes, vk_len, vk, pc, public_input, n_public, flat_proof, proof_len, ok_out, &ext);
}
static int marlin_verify_impl(const uint8_t* vk_bytes, size_t vk_len, const mh_verifier_key* vk, int pc, const uint64_t* public_input, size_t n_public,
                              const uint8_t* flat_proof, size_t proof_len, int* ok_out, const fsh::ExternalFs* ext_fs) {
  if (!vk_bytes || !vk || !flat_proof || !ok_out || (!public_input && n_public)) return fail(MH_EINVAL, "mh_marlin_verify: null pointer");
  if (pc != 0 && pc != 1) return fail(MH_EINVAL, "mh_marlin_verify: pc must be 0 (MarlinKZG10) or 1 (SonicKZG10)");
  if (!vk->g_xy || !vk->gamma_g_xy || !vk->h_xy || !vk->beta_h_xy || !vk->shift_power_h_xy || !vk->shift_power_k_xy)
    return fail(MH_EINVAL, "mh_marlin_verify: null verifier-key element");
  hostverify::VerifierKey k;
  k.pc = pc;
  bool ok_key = load_g1(vk->g_xy, &k.g) && load_g1(vk->gamma_g_xy, &k.gamma_g) && load_g2(vk->h_xy, &k.h) && load_g2(vk->beta_h_xy, &k.beta_h);
  if (pc == 0) ok_key = ok_key && load_g1(vk->shift_power_h_xy, &k.shift_h) && load_g1(vk->shift_power_k_xy, &k.shift_k);
  else ok_key = ok_key && load_g2(vk->shift_power_h_xy, &k.neg_h) && load_g2(vk->shift_power_k_xy, &k.neg_k);
  if (!ok_key) return fail(MH_EINVAL, "mh_marlin_verify: a verifier-key element is not on its curve");
  // G2 has a cofactor on both curves: the pairing is only defined on the order-r subgroup (ark-serialize checks this when a key is read)
  if (!hostpair::g2_in_subgroup(k.h) || !hostpair::g2_in_subgroup(k.beta_h) ||
      (pc == 1 && (!hostpair::g2_in_subgroup(k.neg_h) || !hostpair::g2_in_subgroup(k.neg_k))))
    return fail(MH_EINVAL, "mh_marlin_verify: a G2 element of the verifier key is outside the order-r subgroup");
  std::vector<HFr> pub(n_public);
  for (size_t i = 0; i < n_public; i++) {
    memcpy(pub[i].v, public_input + 4 * i, 32);
    if (HFr::geq_mod(pub[i].v)) return fail(MH_EINVAL, "mh_marlin_verify: public input out of range");
  }
  bool ok = false;
  std::string err;
  if (hostverify::marlin_verify(vk_bytes, vk_len, k, pub, flat_proof, proof_len, &ok, &err, ext_fs) != 0) return fail(MH_EINVAL, "mh_marlin_verify: " + err);
  *ok_out = ok ? 1 : 0;
  return MH_OK;
}
int mh_pairing_product_is_one(const uint64_t* g1_xy, const uint64_t* g2_xy, size_t n, int* is_one_out) {
  if ((!g1_xy || !g2_xy) && n) return fail(MH_EINVAL, "mh_pairing_product_is_one: null pointer");
  if (!is_one_out) return fail(MH_EINVAL, "mh_pairing_product_is_one: null pointer");
  std::vector<hostff::HG1Affine> ps(n);
  std::vector<hostpair::G2Aff> qs(n);
  for (size_t i = 0; i < n; i++)
    if (!load_g1(g1_xy + AFF_L * i, &ps[i]) || !load_g2(g2_xy + 4 * FQ_L * i, &qs[i])) return fail(MH_EINVAL, "mh_pairing_product_is_one: point not on its curve");
  *is_one_out = hostpair::pairing_product_is_one(ps.data(), qs.data(), n) ? 1 : 0;
  return MH_OK;
}

// ---- wire format (host only; wire_host.h) ------------------------------------------------------------------
static int wire_copy_out(const std::vector<uint8_t>& v, uint8_t* out, size_t cap, size_t* len_out, const char* what) {
  if (len_out) *len_out = v.size();
  if (!out) return MH_OK;
  if (cap < v.size()) return fail(MH_EINVAL, std::string(what) + ": buffer too small");
  memcpy(out, v.data(), v.size());
  return MH_OK;
}
int mh_marlin_proof_serialize(const uint8_t* flat, size_t flat_len, int pc, uint8_t* out, size_t cap, size_t* len_out) {
  if (!flat || (pc != 0 && pc != 1)) return fail(MH_EINVAL, "mh_marlin_proof_serialize: bad argument");
  std::vector<uint8_t> v;
  if (!wire::serialize(flat, flat_len, pc, v)) return fail(MH_EINVAL, "mh_marlin_proof_serialize: not a flat proof of this scheme");
  return wire_copy_out(v, out, cap, len_out, "mh_marlin_proof_serialize");
}
int mh_marlin_proof_deserialize(const uint8_t* bytes, size_t len, int pc, uint8_t* flat_out, size_t cap, size_t* len_out) {
  if (!bytes || (pc != 0 && pc != 1)) return fail(MH_EINVAL, "mh_marlin_proof_deserialize: bad argument");
  std::vector<uint8_t> v;
  if (!wire::deserialize(bytes, len, pc, v)) return fail(MH_EINVAL, "mh_marlin_proof_deserialize: invalid encoding (SerializationError::InvalidData)");
  return wire_copy_out(v, flat_out, cap, len_out, "mh_marlin_proof_deserialize");
}
// host-only hook: runs the registered all_gather once (used by the CPU gloo test of the callback plumbing)
int mh_marlin_probe_allgather(const void* send, size_t bytes, void* recv) {
  if (!g_shard.cb) return fail(MH_EINVAL, "no all_gather callback registered");
  return g_shard.cb(send, bytes, recv, g_shard.user);
}
// the registered device exchanges, run once on the caller's device buffers and drained (self-tests of a transport: what the
// sliced sections would call, without a proof around it).  which = 0: all-to-all (bytes = per peer), 1: device all-gather
int mh_marlin_probe_exchange_dev(int which, const void* d_send, size_t bytes, void* d_recv) {
  LOCKED_CTX();
  if (!d_send || !d_recv) return fail(MH_EINVAL, "mh_marlin_probe_exchange_dev: null pointer");
  if (which == 0) {
    if (!g_shard.a2a) return fail(MH_EINVAL, "no all_to_all registered");
    if (!g_shard.a2a_ordered) MH_HIP(hipStreamSynchronize(c.stream));
    ExchangeScope xs(c);
    if (g_shard.a2a(d_send, bytes, d_recv, g_shard.a2a_user) != 0) return fail(MH_EHIP, "all_to_all failed: " + g_err);
  } else {
    if (!g_shard.ag_dev) return fail(MH_EINVAL, "no device all_gather registered");
    if (!g_shard.a2a_ordered) MH_HIP(hipStreamSynchronize(c.stream));
    ExchangeScope xs(c);
    if (g_shard.ag_dev(d_send, bytes, d_recv, g_shard.ag_dev_user) != 0) return fail(MH_EHIP, "device all_gather failed: " + g_err);
  }
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

int mh_marlin_pk_free(uint64_t pk_handle) {
  LOCKED_CTX();
  auto it = g_pks.find(pk_handle);
  if (it == g_pks.end()) return fail(MH_EINVAL, "mh_marlin_pk_free: unknown handle");
  MH_HIP(hipStreamSynchronize(c.stream));
  it->second->free_all();
  g_pks.erase(it);
  return MH_OK;
}

// Marlin::index (lib.rs:100-148)
int mh_marlin_index_pc(const mh_r1cs_matrices* m, uint64_t srs_g, uint64_t srs_gamma_g, int pc, uint64_t* pk_out);
int mh_marlin_index(const mh_r1cs_matrices* m, uint64_t srs_g, uint64_t srs_gamma_g, uint64_t* pk_out) {
  return mh_marlin_index_pc(m, srs_g, srs_gamma_g, 0, pk_out);
}

int mh_marlin_index_pc(const mh_r1cs_matrices* m, uint64_t srs_g, uint64_t srs_gamma_g, int pc, uint64_t* pk_out) {
  LOCKED_CTX();
  if (pc != 0 && pc != 1) return fail(MH_EINVAL, "mh_marlin_index: pc must be 0 (MarlinKZG10) or 1 (SonicKZG10)");
  if (!m || !pk_out) return fail(MH_EINVAL, "mh_marlin_index: null pointer");
  const uint64_t nc = m->num_constraints, ni = m->num_instance;
  if (nc == 0 || ni == 0 || (ni & (ni - 1)) != 0 || ni > nc)
    return fail(MH_EINVAL, "mh_marlin_index: num_instance must be a power of two <= num_constraints (InvalidPublicInputLength)");
  for (int k = 0; k < 3; k++)
    if (!m->row_ptr[k] || (m->row_ptr[k][nc] && !m->col[k])) return fail(MH_EINVAL, "mh_marlin_index: null matrix arrays");
  auto sg = c.bases.find(srs_g), sgg = c.bases.find(srs_gamma_g);
  if (sg == c.bases.end() || sgg == c.bases.end()) return fail(MH_EINVAL, "mh_marlin_index: unknown SRS handle");
  if (sgg->second.n < 3) return fail(MH_EINVAL, "mh_marlin_index: powers_of_gamma_g needs >= 3 points");
  // every commitment of this key multiplies the same powers_of_g: precompute their window table once (msm_fb.cuh).
  // MH_FB=0 keeps the variable-base path.
  {
    static const bool fb_on = [] { const char* e = getenv("MH_FB"); return !e || atoi(e) != 0; }();
    if (fb_on && !sg->second.d_table) {
      int rc = bases_precompute(c, sg->second, 0);
      if (rc != MH_OK && rc != MH_ENOMEM) return rc;      // no room for the table: the variable-base path serves this key
    }
  }

  Trace trace(c);                              // MH_TRACE=1: wall time of every phase on stderr (labels: SURVEY.md Appendix F)
  std::unique_ptr<ProverKey> pkp(new ProverKey());
  ProverKey& pk = *pkp;
  pk.nc = nc; pk.ni = ni; pk.pc = pc;
  pk.srs_g = srs_g; pk.srs_max_degree = sg->second.n - 1;
  {
    uint64_t gg[3 * AFF_L];
    MH_HIP(hipMemcpy(gg, sgg->second.d_points, 3 * PT_B, hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) { memcpy(pk.gamma_g[i].x.v, gg + AFF_L * i, FQ_B); memcpy(pk.gamma_g[i].y.v, gg + AFF_L * i + FQ_L, FQ_B); pk.gamma_g[i].inf = false; }
  }
  // ---- joint matrix (indexer.rs:83-102): per row, sorted union of the column sets --------------------
  std::vector<uint32_t> jcol, jrow;
  {
    const uint64_t total = m->row_ptr[0][nc] + m->row_ptr[1][nc] + m->row_ptr[2][nc];
    jcol.reserve(total); jrow.reserve(total);
    std::vector<uint32_t> cols;
    for (uint64_t r = 0; r < nc; r++) {
      cols.clear();
      for (int k = 0; k < 3; k++)
        for (uint64_t e = m->row_ptr[k][r]; e < m->row_ptr[k][r + 1]; e++) {
          if (m->col[k][e] >= nc) return fail(MH_EINVAL, "mh_marlin_index: column index out of range (NonSquareMatrix)");
          cols.push_back(m->col[k][e]);
        }
      std::sort(cols.begin(), cols.end());
      cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
      jcol.insert(jcol.end(), cols.begin(), cols.end());
      jrow.insert(jrow.end(), cols.size(), (uint32_t)r);
    }
  }
  trace.mark("Marlin::Index: joint matrix (host)");
  pk.nnz = jcol.size();
  pk.H = np2(nc); pk.K = np2(pk.nnz); pk.X = ni;
  pk.logH = log2u(pk.H); pk.logK = log2u(pk.K); pk.logX = log2u(pk.X);
  const uint64_t H = pk.H, K = pk.K, X = pk.X;
  if (pk.logK + 1 > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "PolynomialDegreeTooLarge");
  // AHPForR1CS::max_degree (mod.rs:71-93)
  pk.index_max_degree = std::max(std::max(2 * H + 1 - 2, 3 * H + 2 - 3), std::max(H, K - 1));
  if (pk.srs_max_degree < pk.index_max_degree) return fail(MH_EINVAL, "IndexTooLarge: SRS max degree below the index's");
  if (K - 2 > pk.srs_max_degree) return fail(MH_EINVAL, "IndexTooLarge");
  if (pc == 1) {
    // SonicKZG10::trim: shifted_powers_of_gamma_g[d] = powers_of_gamma_g[max_degree - d + i], i = 0..2
    auto fetch = [&](uint64_t off, HG1Affine* out3) -> int {
      if (off + 3 > sgg->second.n) return fail(MH_EINVAL, "SonicKZG10 needs powers_of_gamma_g up to max_degree + 1");
      uint64_t gg[3 * AFF_L];
      MH_HIP(hipMemcpy(gg, (const char*)sgg->second.d_points + off * PT_B, 3 * PT_B, hipMemcpyDeviceToHost));
      for (int i = 0; i < 3; i++) { memcpy(out3[i].x.v, gg + AFF_L * i, FQ_B); memcpy(out3[i].y.v, gg + AFF_L * i + FQ_L, FQ_B); out3[i].inf = false; }
      return MH_OK;
    };
    MH_TRY(fetch(pk.srs_max_degree - (H - 2), pk.gamma_g_h));
    MH_TRY(fetch(pk.srs_max_degree - (K - 2), pk.gamma_g_k));
  }

  // ---- matrices on the device: CSR A, B stay for z_A, z_B (prover.rs:256-276); C only serves the arithmetisation ----
  Csr* cs[2] = {&pk.A, &pk.B};
  Csr csC;
  for (int q = 0; q < 3; q++) {
    Csr* d = q < 2 ? cs[q] : &csC;
    uint64_t nz = m->row_ptr[q][nc];
    d->nnz = nz;
    MH_TRY(d->row_ptr.alloc((nc + 1) * 8)); MH_TRY(h2d(c, d->row_ptr.p, m->row_ptr[q], (nc + 1) * 8));
    MH_TRY(d->col.alloc(std::max<uint64_t>(nz, 1) * 4)); MH_TRY(h2d(c, d->col.p, m->col[q], nz * 4));
    d->has_val = m->val[q] != nullptr;
    if (d->has_val) { MH_TRY(d->val.alloc(std::max<uint64_t>(nz, 1) * 32)); MH_TRY(h2d(c, d->val.p, m->val[q], nz * 32)); }
  }
  // ---- arithmetize_matrix (constraint_systems.rs:125-262): one device thread per entry of the joint matrix --------
  DBuf* evs[6] = {&pk.ev_row, &pk.ev_col, &pk.ev_val_a, &pk.ev_val_b, &pk.ev_val_c, &pk.ev_row_col};
  DBuf* pls[6] = {&pk.p_row, &pk.p_col, &pk.p_a_val, &pk.p_b_val, &pk.p_c_val, &pk.p_row_col};   // INDEXER_POLYNOMIALS order
  {
    HFr w = hostff::fr_two_adic_root();
    for (uint32_t i = pk.logH; i < hostff::FR_TWO_ADICITY_H; i++) w = w.sqr();
    const HFr h_inv = HFr::from_u64(H).inv();
    DBuf d_jrow, d_jcol;
    MH_TRY(d_jrow.alloc(std::max<uint64_t>(pk.nnz, 1) * 4)); MH_TRY(d_jcol.alloc(std::max<uint64_t>(pk.nnz, 1) * 4));
    MH_TRY(h2d(c, d_jrow.p, jrow.data(), pk.nnz * 4)); MH_TRY(h2d(c, d_jcol.p, jcol.data(), pk.nnz * 4));
    for (int q = 0; q < 6; q++) { MH_TRY(evs[q]->alloc(K * 32)); MH_TRY(pls[q]->alloc(K * 32)); }
    auto mat = [](const Csr& s) { return poly::ArithMat{(const u64*)s.row_ptr.p, (const u32*)s.col.p, s.has_val ? (const Fr*)s.val.p : nullptr}; };
    KLAUNCH(poly::arithmetize_kernel, K, pk.ev_row.fr(), pk.ev_col.fr(), pk.ev_row_col.fr(), pk.ev_val_a.fr(), pk.ev_val_b.fr(), pk.ev_val_c.fr(),
            (const u32*)d_jrow.p, (const u32*)d_jcol.p, (u64)pk.nnz, (u64)K, mat(pk.A), mat(pk.B), mat(csC), arg(w), arg(h_inv), (u64)H, (u64)X);
    MH_HIP(hipGetLastError());
    MH_HIP(hipStreamSynchronize(c.stream));
    d_jrow.release(); d_jcol.release();
    csC.row_ptr.release(); csC.col.release(); csC.val.release();
  }
  trace.mark("Marlin::Index: arithmetize_matrix x3 (device)");
  for (int q = 0; q < 6; q++) MH_TRY(ntt_device(c, evs[q]->p, pls[q]->p, pk.logK, 1));   // interpolate (constraint_systems.rs:234-239)
  trace.mark("Marlin::Index: 6 interpolations over K");
  // ---- the same six polynomials on the coset g K: the third round evaluates h_2 there (see mh_marlin_prove) ------
  {
    HFr g = HFr::from_u64(7);
    while (g.pow_u64(K) == HFr::one()) g = g + HFr::one();
    pk.coset_g = g;
    const uint64_t LO = 1ull << poly::COSET_LO_BITS, nhi = (K + LO - 1) / LO;
    std::vector<uint64_t> lo(4 * LO), hi(4 * nhi), ilo(4 * LO), ihi(4 * nhi);
    const HFr ginv = g.inv(), gL = g.pow_u64(LO), gLinv = ginv.pow_u64(LO);
    HFr a = HFr::one(), b = HFr::one();
    for (uint64_t j = 0; j < LO; j++) { memcpy(&lo[4 * j], a.v, 32); memcpy(&ilo[4 * j], b.v, 32); a = a * g; b = b * ginv; }
    a = HFr::one(); b = HFr::one();
    for (uint64_t j = 0; j < nhi; j++) { memcpy(&hi[4 * j], a.v, 32); memcpy(&ihi[4 * j], b.v, 32); a = a * gL; b = b * gLinv; }
    MH_TRY(pk.gpow_lo.alloc(LO * 32)); MH_TRY(pk.ginv_lo.alloc(LO * 32)); MH_TRY(pk.gpow_hi.alloc(nhi * 32)); MH_TRY(pk.ginv_hi.alloc(nhi * 32));
    MH_TRY(h2d(c, pk.gpow_lo.p, lo.data(), LO * 32)); MH_TRY(h2d(c, pk.ginv_lo.p, ilo.data(), LO * 32));
    MH_TRY(h2d(c, pk.gpow_hi.p, hi.data(), nhi * 32)); MH_TRY(h2d(c, pk.ginv_hi.p, ihi.data(), nhi * 32));
    MH_HIP(hipStreamSynchronize(c.stream));
    DBuf* css[6] = {&pk.cs_row, &pk.cs_col, &pk.cs_val_a, &pk.cs_val_b, &pk.cs_val_c, &pk.cs_row_col};
    DBuf tmp;
    MH_TRY(tmp.alloc(K * 32));
    for (int q = 0; q < 6; q++) {
      int rc = css[q]->alloc(K * 32);
      if (rc != MH_OK) { tmp.release(); return rc; }
      KLAUNCH(poly::twist_kernel, K, tmp.fr(), (const Fr*)pls[q]->p, (const Fr*)pk.gpow_hi.p, (const Fr*)pk.gpow_lo.p, (u64)K);
      rc = ntt_device(c, tmp.p, css[q]->p, pk.logK, 0);
      if (rc != MH_OK) { tmp.release(); return rc; }
    }
    MH_HIP(hipStreamSynchronize(c.stream));
    tmp.release();
  }
  MH_HIP(hipStreamSynchronize(c.stream));
  trace.mark("Marlin::Index: 6 evaluations on the coset g K");
  // ---- index commitments: PC::commit(ck, index.iter(), None) (lib.rs:123-126) ----------------------------
  pk.index_comms.resize(6);
  {
    // one batch: the six MSMs share the sort / accumulate / reduce launches (never sharded: every rank indexes in full)
    const void* b6[6]; const void* s6[6]; size_t n6[6];
    for (int q = 0; q < 6; q++) { b6[q] = sg->second.d_points; s6[q] = pls[q]->p; n6[q] = K; }
    uint64_t xyz[6 * XYZ_L];
    MH_TRY(msm_batch_device(c, 6, b6, s6, n6, 1, xyz));
    HG1 jac[6]; HG1Affine aff[6];
    for (int q = 0; q < 6; q++) jac[q] = jac_from(xyz + XYZ_L * q);
    hostff::batch_to_affine(jac, 6, aff);
    for (int q = 0; q < 6; q++) { pk.index_comms[q].comm = aff[q]; pk.index_comms[q].has_shifted = false; }
  }
  trace.mark("Commit to index polynomials");
  // IndexVerifierKey::write (data_structures.rs:36-43)
  fsh::put_u64(pk.vk_bytes, nc); fsh::put_u64(pk.vk_bytes, nc); fsh::put_u64(pk.vk_bytes, pk.nnz);
  for (auto& cm : pk.index_comms) put_comm(pk.vk_bytes, cm, pk.pc);

  // ---- calculate_t structure (prover.rs:411-428): entries grouped by output index, then by matrix ----------
  {
    uint64_t total = m->row_ptr[0][nc] + m->row_ptr[1][nc] + m->row_ptr[2][nc];
    pk.t_has_coef = m->val[0] || m->val[1] || m->val[2];
    // counting sort by key = (k, matrix)
    std::vector<uint64_t> cnt(3 * H + 1, 0);
    for (int q = 0; q < 3; q++)
      for (uint64_t e = 0; e < m->row_ptr[q][nc]; e++) cnt[3 * reindex_by_subdomain(H, X, m->col[q][e]) + q + 1]++;
    for (uint64_t i = 0; i < 3 * H; i++) cnt[i + 1] += cnt[i];
    std::vector<uint32_t> erow(total);
    std::vector<uint64_t> ecoef(pk.t_has_coef ? total * 4 : 0);
    std::vector<uint64_t> cursor(cnt.begin(), cnt.end() - 1);
    for (int q = 0; q < 3; q++)
      for (uint64_t r = 0; r < nc; r++)
        for (uint64_t e = m->row_ptr[q][r]; e < m->row_ptr[q][r + 1]; e++) {
          uint64_t key = 3 * reindex_by_subdomain(H, X, m->col[q][e]) + q;
          uint64_t pos = cursor[key]++;
          erow[pos] = (uint32_t)r;
          if (pk.t_has_coef) { if (m->val[q]) memcpy(&ecoef[4 * pos], m->val[q] + 4 * e, 32); else memcpy(&ecoef[4 * pos], HFr::one().v, 32); }
        }
    std::vector<poly::TItem> items;
    // two-level grouping: items (<= 128 entries) -> groups (<= 128 items) -> output k
    std::vector<uint64_t> group_ptr, item_ptr(H + 1, 0);     // group_ptr: item ranges; item_ptr: group ranges per k
    const uint32_t ITEM = 128, GROUP = 128;
    for (uint64_t k = 0; k < H; k++) {
      item_ptr[k] = group_ptr.size();
      uint64_t first_item = items.size();
      for (int q = 0; q < 3; q++) {
        uint64_t lo = cnt[3 * k + q], hi = cnt[3 * k + q + 1];
        for (uint64_t s = lo; s < hi; s += ITEM) {
          poly::TItem it; it.first = s; it.count = (uint32_t)std::min<uint64_t>(ITEM, hi - s); it.k = (uint32_t)k; it.m = q; it.pad = 0;
          items.push_back(it);
        }
      }
      for (uint64_t g = first_item; g < items.size(); g += GROUP) group_ptr.push_back(g);
    }
    item_ptr[H] = group_ptr.size();
    pk.t_ngroups = group_ptr.size();
    group_ptr.push_back(items.size());
    pk.t_nitems = items.size();
    MH_TRY(pk.t_items.alloc(items.size() * sizeof(poly::TItem))); MH_TRY(h2d(c, pk.t_items.p, items.data(), items.size() * sizeof(poly::TItem)));
    MH_TRY(pk.t_erow.alloc(total * 4)); MH_TRY(h2d(c, pk.t_erow.p, erow.data(), total * 4));
    if (pk.t_has_coef) { MH_TRY(pk.t_ecoef.alloc(total * 32)); MH_TRY(h2d(c, pk.t_ecoef.p, ecoef.data(), total * 32)); }
    MH_TRY(pk.t_item_ptr.alloc((H + 1) * 8)); MH_TRY(h2d(c, pk.t_item_ptr.p, item_ptr.data(), (H + 1) * 8));
    MH_TRY(pk.t_group_ptr.alloc(group_ptr.size() * 8)); MH_TRY(h2d(c, pk.t_group_ptr.p, group_ptr.data(), group_ptr.size() * 8));
    MH_HIP(hipStreamSynchronize(c.stream));
  }
  trace.mark("Marlin::Index: calculate_t grouping (host)");
  // ---- workspace ------------------------------------------------------------------------------------------------
  const uint64_t big = std::max<uint64_t>(2 * K, 4 * H) + 64;
  MH_TRY(pk.z.alloc(H * 32)); MH_TRY(pk.za_ev.alloc(H * 32)); MH_TRY(pk.zb_ev.alloc(H * 32)); MH_TRY(pk.xpoly.alloc((X + 8) * 32));
  MH_TRY(pk.w.alloc((H + 8) * 32)); MH_TRY(pk.za.alloc((H + 8) * 32)); MH_TRY(pk.zb.alloc((H + 8) * 32));
  MH_TRY(pk.mask.alloc((3 * H + 8) * 32)); MH_TRY(pk.t.alloc(H * 32)); MH_TRY(pk.g1.alloc(H * 32)); MH_TRY(pk.h1.alloc((3 * H + 8) * 32));
  MH_TRY(pk.g2.alloc(K * 32)); MH_TRY(pk.h2.alloc(K * 32)); MH_TRY(pk.outer.alloc((3 * H + 8) * 32)); MH_TRY(pk.inner.alloc(K * 32));
  for (auto& s : pk.S) MH_TRY(s.alloc(big * 32));
  MH_TRY(pk.small.alloc((big / 8 + pk.t_nitems + pk.t_ngroups + 4096) * 32));
  MH_TRY(pk.scal.alloc(256));
  MH_TRY(ensure_twiddles_public(c, std::max(pk.logK + 1, pk.logH + 2)));
  MH_HIP(hipStreamSynchronize(c.stream));
  trace.mark("Marlin::Index: workspace + twiddles");
  uint64_t h = mh::g_next_handle++;
  g_pks[h] = std::move(pkp);
  *pk_out = h;
  return MH_OK;
}

int mh_marlin_pk_info(uint64_t pk_handle, uint64_t* info8) {
  LOCKED_CTX();
  auto it = g_pks.find(pk_handle);
  if (it == g_pks.end() || !info8) return fail(MH_EINVAL, "mh_marlin_pk_info: bad argument");
  ProverKey& pk = *it->second;
  info8[0] = pk.H; info8[1] = pk.K; info8[2] = pk.X; info8[3] = pk.nnz; info8[4] = pk.index_max_degree;
  info8[5] = pk.srs_max_degree; info8[6] = pk.nc; info8[7] = pk.ni;
  return MH_OK;
}

int mh_marlin_vk_bytes(uint64_t pk_handle, uint8_t* out, size_t cap, size_t* len_out) {
  LOCKED_CTX();
  auto it = g_pks.find(pk_handle);
  if (it == g_pks.end()) return fail(MH_EINVAL, "mh_marlin_vk_bytes: unknown handle");
  if (len_out) *len_out = it->second->vk_bytes.size();
  if (out) {
    if (cap < it->second->vk_bytes.size()) return fail(MH_EINVAL, "mh_marlin_vk_bytes: buffer too small");
    memcpy(out, it->second->vk_bytes.data(), it->second->vk_bytes.size());
  }
  return MH_OK;
}

// Coefficients (Montgomery) of a prover / indexer polynomial of the last proof, by the reference's label
// (PROVER_POLYNOMIALS / INDEXER_POLYNOMIALS, src/ahp/mod.rs:33-45).  out == NULL queries the length.
int mh_marlin_get_poly(uint64_t pk_handle, const char* label, uint64_t* out, size_t cap_elems, size_t* len_out) {
  LOCKED_CTX();
  auto it = g_pks.find(pk_handle);
  if (it == g_pks.end() || !label) return fail(MH_EINVAL, "mh_marlin_get_poly: bad argument");
  auto pit = it->second->last_polys.find(label);
  if (pit == it->second->last_polys.end()) return fail(MH_EINVAL, "mh_marlin_get_poly: unknown label (or no proof yet)");
  if (len_out) *len_out = pit->second.second;
  if (!out) return MH_OK;
  if (cap_elems < pit->second.second) return fail(MH_EINVAL, "mh_marlin_get_poly: buffer too small");
  MH_HIP(hipMemcpyAsync(out, pit->second.first, pit->second.second * 32, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

// Marlin::prove (lib.rs:151-311).  instance: formatted public input (X elements, leading one included);
// witness: nc - X elements (padding witnesses included).  zk_rng = ChaCha(zk_seed, rounds) drawn in the
// order of SURVEY.md Appendix C.  proof_out: the flat ToBytes-layout proof.
static int marlin_prove_impl(uint64_t pk_handle, const uint64_t* instance, const uint64_t* witness, bool inputs_on_device,
                             const uint8_t* zk_seed, int zk_rounds, const uint64_t* zk_draws, size_t n_draws, uint8_t* proof_out, size_t cap,
                             size_t* len_out, const fsh::ExternalFs* ext_fs = nullptr);
int mh_marlin_prove(uint64_t pk_handle, const uint64_t* instance, const uint64_t* witness, const uint8_t* zk_seed,
                    int zk_rounds, uint8_t* proof_out, size_t cap, size_t* len_out) {
  return marlin_prove_impl(pk_handle, instance, witness, false, zk_seed, zk_rounds, nullptr, 0, proof_out, cap, len_out);
}
// the same with the assignment already in device memory (a witness generator that runs on the GPU, or a caller that
// uploaded it ahead of time): no PCIe transfer inside the call
int mh_marlin_prove_dev(uint64_t pk_handle, const void* d_instance, const void* d_witness, const uint8_t* zk_seed,
                        int zk_rounds, uint8_t* proof_out, size_t cap, size_t* len_out) {
  return marlin_prove_impl(pk_handle, (const uint64_t*)d_instance, (const uint64_t*)d_witness, true, zk_seed, zk_rounds, nullptr, 0, proof_out, cap, len_out);
}
// Marlin::prove for a caller whose `zk_rng` is not a ChaCha generator (src/lib.rs:151-155 is generic over R: RngCore): the
// caller draws the field elements itself -- `F::rand(zk_rng)`, mh_marlin_zk_draw_count of them, in the order the reference
// consumes them -- and hands them over (host memory, Montgomery limbs).
int mh_marlin_zk_draw_count(uint64_t pk_handle, size_t* n_out) {
  LOCKED_CTX();
  auto it = g_pks.find(pk_handle);
  if (it == g_pks.end() || !n_out) return fail(MH_EINVAL, "mh_marlin_zk_draw_count: bad argument");
  // r_w, r_za, r_zb | 3|H| mask coefficients | 3 each for w, z_a, z_b, g_1 (+ g_1's shifted commitment with MarlinKZG10)
  *n_out = 3 + 3 * it->second->H + (it->second->pc == 1 ? 12 : 15);
  return MH_OK;
}
int mh_marlin_prove_draws(uint64_t pk_handle, const uint64_t* instance, const uint64_t* witness, const uint64_t* zk_draws, size_t n_draws,
                          uint8_t* proof_out, size_t cap, size_t* len_out) {
  if (!zk_draws) return fail(MH_EINVAL, "mh_marlin_prove_draws: null draws");
  static const uint8_t no_seed[32] = {0};
  return marlin_prove_impl(pk_handle, instance, witness, false, no_seed, 20, zk_draws, n_draws, proof_out, cap, len_out);
}
// Marlin<F, PC, FS>::prove for ANY `FS: FiatShamirRng` (src/lib.rs:64-70,151-155): the transcript operations go to the caller's
// callbacks -- initialize (src/lib.rs:161-163), absorb after every round's commitments and after the evaluations
// (:180,:201,:221,:289), and RngCore::next_u64 under `F::rand(fs_rng)` / `u128::rand(fs_rng)` (src/ahp/verifier.rs:58-61,86,98;
// src/lib.rs:290).  With a sharded prove every rank's callbacks see the same byte sequence.
static bool fs_callbacks_ok(const mh_fiat_shamir* fs) { return fs && fs->initialize && fs->absorb && fs->next_u64; }
int mh_marlin_prove_fs(uint64_t pk_handle, const uint64_t* instance, const uint64_t* witness, const uint8_t* zk_seed, int zk_rounds,
                       const mh_fiat_shamir* fs, uint8_t* proof_out, size_t cap, size_t* len_out) {
  if (!fs_callbacks_ok(fs)) return fail(MH_EINVAL, "mh_marlin_prove_fs: initialize, absorb and next_u64 must all be set");
  static_assert(sizeof(mh_fiat_shamir) == sizeof(fsh::ExternalFs), "mh_fiat_shamir and fsh::ExternalFs are the same struct");
  const fsh::ExternalFs ext{fs->user, fs->initialize, fs->absorb, fs->next_u64};
  return marlin_prove_impl(pk_handle, instance, witness, false, zk_seed, zk_rounds, nullptr, 0, proof_out, cap, len_out, &ext);
}
// inside the prover a poisoned rank (JobStatus) launches nothing more: its buffers may be the ones that could not be set up
#undef KLAUNCH
#define KLAUNCH(kern, n, ...)                                                                          \
  do {                                                                                                 \
    if (g_job.poison == MH_OK) hipLaunchKernelGGL(kern, dim3(poly::grid_for(n)), dim3(poly::TPB), 0, c.stream, __VA_ARGS__); \
  } while (0)
static int marlin_prove_impl(uint64_t pk_handle, const uint64_t* instance, const uint64_t* witness, bool inputs_on_device,
                             const uint8_t* zk_seed, int zk_rounds, const uint64_t* zk_draws, size_t n_draws, uint8_t* proof_out, size_t cap,
                             size_t* len_out, const fsh::ExternalFs* ext_fs) {
  LOCKED_CTX();
  auto pit = g_pks.find(pk_handle);
  if (pit == g_pks.end()) return fail(MH_EINVAL, "mh_marlin_prove: unknown prover key");
  if (!instance || !witness || !zk_seed || !proof_out) return fail(MH_EINVAL, "mh_marlin_prove: null pointer");
  if (zk_rounds != 8 && zk_rounds != 12 && zk_rounds != 20) return fail(MH_EINVAL, "mh_marlin_prove: ChaCha rounds must be 8, 12 or 20");
  ProverKey& pk = *pit->second;
  const uint64_t H = pk.H, K = pk.K, X = pk.X, nc = pk.nc;
  const uint64_t nw = nc - X;
  const uint32_t lgH = pk.logH, lgK = pk.logK, lgX = pk.logX;
  ZkSource zk;
  if (zk_draws) { zk.draws = zk_draws; zk.n = n_draws; } else zk.chacha = fsh::ChaChaRng(zk_seed, zk_rounds);
  Trace tr(c);
  Fr* S[8]; for (int i = 0; i < 8; i++) S[i] = pk.S[i].fr();
  g_job = JobStatus();                          // see JobStatus: a rank that fails locally fails the job instead of hanging it
  // ... and whichever way the proof ends, nothing of its status outlives it: the building blocks below are also reachable through
  // mh_msm_batch_sharded_dev / mh_ntt_dist_dev, and a stale `failed` or `poison` there would fail or silently skip work
  struct JobGuard { ~JobGuard() { g_job = JobStatus(); } } job_guard;
  g_job.buf[0] = S[6]; g_job.buf[1] = S[7]; g_job.buf_bytes = std::min(pk.S[6].bytes, pk.S[7].bytes);
  const Fr* tw = (const Fr*)c.tw;
  // sliced sections (rounds 2 and 3 on 1 / G of every 4H- and K-sized vector): with an all-to-all registered and a
  // power-of-two number of ranks; MH_SLICED=0 keeps every rank on the replicated rounds
  static const bool sliced_env = [] { const char* e = getenv("MH_SLICED"); return !e || atoi(e) != 0; }();
  const uint64_t Gs = (uint64_t)g_shard.world;
  // (from 4 ranks on: with 2 ranks each exchange moves a quarter of the vector to the one peer -- ~10 ms of xGMI time per proof
  // at 2^20 for 2.4 ms of saved kernels; MH_SLICED=2 forces it for tests)
  static const bool sliced_force = [] { const char* e = getenv("MH_SLICED"); return e && atoi(e) == 2; }();
  const bool sliced = sliced_env && (Gs >= 4 || (sliced_force && Gs > 1)) && g_shard.a2a && (Gs & (Gs - 1)) == 0 && Gs <= 8 && H >= Gs * Gs;
  if (sliced) PTRY(prepare_sliced(c, pk));

  // ---------------- prover_init (prover.rs:211-306): z = x || w, z_A = A z, z_B = B z --------------------
  if (inputs_on_device) { PTRY(d2d(c, pk.z.fr(), (const Fr*)instance, X)); PTRY(d2d(c, pk.z.fr() + X, (const Fr*)witness, nw)); }
  else { PTRY(h2d(c, pk.z.fr(), instance, X * 32)); PTRY(h2d(c, pk.z.fr() + X, witness, nw * 32)); }
  {
    ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::spmv_kernel, nc, pk.za_ev.fr(), (const u64*)pk.A.row_ptr.p, (const u32*)pk.A.col.p,
            pk.A.has_val ? (const Fr*)pk.A.val.p : (const Fr*)nullptr, (const Fr*)pk.z.fr(), (u64)nc);
    KLAUNCH(poly::spmv_kernel, nc, pk.zb_ev.fr(), (const u64*)pk.B.row_ptr.p, (const u32*)pk.B.col.p,
            pk.B.has_val ? (const Fr*)pk.B.val.p : (const Fr*)nullptr, (const Fr*)pk.z.fr(), (u64)nc);
  }
  PTRY(zero_tail(c, pk.za_ev.fr(), nc, H)); PTRY(zero_tail(c, pk.zb_ev.fr(), nc, H));
  std::vector<HFr> pub(X - 1);                      // public_input() = formatted input without the leading one
  for (uint64_t i = 1; i < X; i++) memcpy(pub[i - 1].v, instance + 4 * i, 32);
  fsh::FiatShamirRng fs;
  fs.ext = ext_fs;                                        // the caller's FS: FiatShamirRng, or SimpleHashFiatShamirRng<Blake2s, ChaChaRng>
  {
    std::vector<uint8_t> init; const char* name = "MARLIN-2019";
    init.insert(init.end(), name, name + 11);
    init.insert(init.end(), pk.vk_bytes.begin(), pk.vk_bytes.end());
    for (auto& x : pub) fsh::put_fr(init, x);
    fs.initialize(init);                                  // lib.rs:161-163
  }

  tr.mark("AHP::Prover::Init (z_A, z_B)"); mh::host_tick("init issued");
  // ---------------- first round (prover.rs:309-409) -----------------------------------------------------------
  PTRY(ntt_device(c, pk.z.fr(), pk.xpoly.fr(), lgX, 1));              // x_poly = interpolate(formatted input)
  if (X <= 16) {                                                        // x_evals = domain_h.fft(x_poly): Horner per point
    ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::eval_small_poly_kernel, H, S[1], (const Fr*)pk.xpoly.fr(), (u32)X, tw, lgH);
  } else {
    PTRY(d2d(c, S[0], pk.xpoly.fr(), X)); PTRY(zero_tail(c, S[0], X, H));
    PTRY(ntt_device(c, S[0], S[1], lgH, 0));
  }
  { ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::w_evals_kernel, H, S[2], (const Fr*)(pk.z.fr() + X), (u64)nw, (const Fr*)S[1], (u64)H, (u64)(H / X)); }
  PTRY(ntt_device(c, S[2], S[3], lgH, 1));
  // + r * v_H: the reference multiplies by FFT (prover.rs:352); the product is exactly [-r, 0.., 0, r]
  HFr r_w = zk.fr();
  if (g_job.poison == MH_OK) hipLaunchKernelGGL(poly::blind_vanishing_kernel, dim3(1), dim3(64), 0, c.stream, S[3], (u64)H, arg(r_w));
  const uint64_t w_len = H + 1 - X;                                       // (w + r v_H) / v_X, remainder zero
  PTRY(div_vanishing(c, pk.w.fr(), S[3], H + 1, X, S[4]));
  auto blind_h = [&](Fr* dst, const Fr* evals, HFr* r_out) -> int {
    PTRY(ntt_device(c, evals, dst, lgH, 1));
    HFr r = zk.fr(); *r_out = r;
    if (g_job.poison == MH_OK) hipLaunchKernelGGL(poly::blind_vanishing_kernel, dim3(1), dim3(64), 0, c.stream, dst, (u64)H, arg(r));
    return MH_OK;
  };
  HFr r_za, r_zb;
  PTRY(blind_h(pk.za.fr(), pk.za_ev.fr(), &r_za));
  PTRY(blind_h(pk.zb.fr(), pk.zb_ev.fr(), &r_zb));
  const uint64_t za_len = H + 1;
  // mask polynomial (prover.rs:369-381): 3H sequential draws, then force sum over H to zero
  const uint64_t mask_len = 3 * H;          // degree 3H + 2*zk_bound - 3
  PTRY(device_poly_rand(c, pk, zk, pk.mask.fr(), mask_len, S[0], (u32*)S[1]));
  if (g_job.poison == MH_OK) hipLaunchKernelGGL(rng::mask_fix_kernel, dim3(1), dim3(1), 0, c.stream, pk.mask.fr(), (u64)H, (u64)mask_len);
  tr.mark("AHP::Prover::FirstRound (w, z_A, z_B, mask polys)"); mh::host_tick("round 1 kernels issued");
  // Three of round 2's five forward 4H transforms need no challenge: z_a and z_b (prover.rs:467) and z = w v_X + x
  // (prover.rs:503-516, 534).  They are handed to the commitment's MSM batch as its side job (Context::side_job): issued on the
  // second stream behind the bucket accumulation, they run beside the bucket reduction -- one wave per SIMD, a third of the
  // issue slots idle -- and through the host round trip that brings the commitments back.  Same values, same buffers (S[0],
  // S[1], S[7] -> S[2]; nothing else touches them before round 2).  Not with sliced rounds (the transforms are distributed
  // there).
  const uint32_t lg4H = lgH + 2; const uint64_t H4 = 4 * H;
  const uint64_t z_len = w_len + X;
  const bool side_ntt = !sliced;
  auto early_transforms = [&]() -> int {
    PTRY(ntt_device_len(c, pk.za.fr(), za_len, S[0], lg4H, 0));
    PTRY(ntt_device_len(c, pk.zb.fr(), za_len, S[1], lg4H, 0));
    { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::z_poly_kernel, z_len, S[7], (const Fr*)pk.w.fr(), (u64)w_len, (u64)X, (const Fr*)pk.xpoly.fr(), (u64)X); }
    PTRY(ntt_device_len(c, S[7], z_len, S[2], lg4H, 0));
    return MH_OK;
  };
  struct SideJobGuard { Context& c; ~SideJobGuard() { c.side_job = nullptr; } } side_guard{c};   // the job captures this frame
  c.side_ran = false;
  if (side_ntt) c.side_job = early_transforms;
  // PC::commit first round (lib.rs:172): w, z_a, z_b hiding 1; mask none
  std::vector<fsh::Commitment> cm1; std::vector<PolyRand> rd1;
  CTRY(marlin_commit(c, pk, {{pk.w.fr(), w_len, false, 0, true}, {pk.za.fr(), za_len, false, 0, true},
                               {pk.zb.fr(), za_len, false, 0, true}, {pk.mask.fr(), mask_len, false, 0, false}}, &zk, cm1, rd1));
  if (side_ntt) {
    if (c.side_ran) { PHIP(hipStreamWaitEvent(c.stream, c.side_ev[1], 0)); c.side_ran = false; }   // round 2 starts behind the side stream
    else { c.side_job = nullptr; PTRY(early_transforms()); }                                       // the batch never took it (variable-base path)
  }
  fsh::Commitment &c_w = cm1[0], &c_za = cm1[1], &c_zb = cm1[2], &c_mask = cm1[3];
  PolyRand &rd_w = rd1[0], &rd_za = rd1[1], &rd_zb = rd1[2];
  {
    std::vector<uint8_t> b;
    put_comm(b, c_w, pk.pc); put_comm(b, c_za, pk.pc); put_comm(b, c_zb, pk.pc); put_comm(b, c_mask, pk.pc);
    fs.absorb(b);                                                            // lib.rs:180
  }
  tr.mark("Committing to first round polys"); mh::host_tick("round 1 committed + absorbed");
  // verifier_first_round (verifier.rs:44-79)
  auto v_h = [&](const HFr& x) { return x.pow_u64(H) - HFr::one(); };
  HFr alpha = fs.rand_fr();
  while (v_h(alpha).is_zero()) alpha = fs.rand_fr();
  HFr eta_a = fs.rand_fr(), eta_b = fs.rand_fr(), eta_c = fs.rand_fr();

  // ---------------- second round (prover.rs:443-570) ---------------------------------------------------------
  const uint64_t H4loc = sliced ? H4 / Gs : H4;          // elements of a 4H-vector this rank works on
  if (sliced) {
    CTRY(ntt_dist_device(c, S[0], slice_c(c, S[0], pk.za.fr(), za_len), S[0], lg4H, 0));
    CTRY(ntt_dist_device(c, S[1], slice_c(c, S[1], pk.zb.fr(), za_len), S[1], lg4H, 0));
  } else if (!side_ntt) {
  PTRY(ntt_device_len(c, pk.za.fr(), za_len, S[0], lg4H, 0));
  PTRY(ntt_device_len(c, pk.zb.fr(), za_len, S[1], lg4H, 0));
  }
  // The reference forms z_c = z_a * z_b (two forward transforms, a pointwise product, one inverse: prover.rs:467),
  // summed_z_m = eta_a z_a + eta_b z_b + eta_c z_c (468-471), and later evaluates summed_z_m on the same 4H domain
  // (533).  deg z_c = 2H + 2 < 4H, so those evaluations ARE eta_c z_a z_b + eta_a z_a + eta_b z_b pointwise: the inverse
  // and the forward transform in between cancel exactly and are not executed.  S[1] <- evaluations of summed_z_m.
  { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::summed_evals_kernel, H4loc, S[1], (const Fr*)S[0], arg(eta_a), arg(eta_b), arg(eta_c), (u64)H4loc); }
  // r_alpha_x evals on H (mod.rs:311-318)
  HFr vH_alpha = v_h(alpha);
  { ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::x_minus_elements_kernel, H, S[4], tw, arg(alpha), lgH);
  }
  PTRY(batch_inverse(c, S[4], S[5], H, vH_alpha, 1));
  PTRY(ntt_device(c, S[4], S[5], lgH, 1));                                 // r_alpha_poly
  // t (prover.rs:411-428)
  { ProfScope ps(c, PF_GLUE);
    Fr* partial = pk.small.fr();
    KLAUNCH(poly::t_items_kernel, pk.t_nitems, partial, (const poly::TItem*)pk.t_items.p, (u64)pk.t_nitems, (const u32*)pk.t_erow.p,
            pk.t_has_coef ? (const Fr*)pk.t_ecoef.p : (const Fr*)nullptr, (const Fr*)S[4], arg(eta_a), arg(eta_b), arg(eta_c));
    Fr* partial2 = partial + pk.t_nitems;
    KLAUNCH(poly::t_sum_kernel, pk.t_ngroups, partial2, (const Fr*)partial, (const u64*)pk.t_group_ptr.p, (u64)pk.t_ngroups);
    KLAUNCH(poly::t_sum_kernel, H, S[6], (const Fr*)partial2, (const u64*)pk.t_item_ptr.p, (u64)H); }
  PTRY(ntt_device(c, S[6], pk.t.fr(), lgH, 1));
  // z = w * v_X + x  (prover.rs:503-516)
  if (!side_ntt) { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::z_poly_kernel, z_len, S[7], (const Fr*)pk.w.fr(), (u64)w_len, (u64)X, (const Fr*)pk.xpoly.fr(), (u64)X); }
  // q_1 (prover.rs:520-547): forward transforms on the 4H domain (summed_z_m's is known, see above), pointwise, one inverse
  const uint64_t h1_len = 2 * H;                                                // deg <= 2H + 2*zk_bound - 2 - ... (upper H coefficients are zero)
  if (sliced) {
    // the same on this rank's slices: C-layout slices of r_alpha, z, t -> distributed forward transforms -> pointwise on its
    // M-layout block -> distributed inverse -> q_1, the division by v_H (stride H / G inside a slice) and the remainder, all
    // local; then ONE all-gather returns h_1 and x g_1 to every rank
    const uint64_t Hloc = H / Gs;
    CTRY(ntt_dist_device(c, S[0], slice_c(c, S[0], S[5], H), S[0], lg4H, 0));                  // r_alpha
    CTRY(ntt_dist_device(c, S[2], slice_c(c, S[2], S[7], z_len), S[2], lg4H, 0));              // z
    CTRY(ntt_dist_device(c, S[4], slice_c(c, S[4], pk.t.fr(), H), S[4], lg4H, 0));             // t
    { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::mul_sub_mul_kernel, H4loc, S[0], (const Fr*)S[0], (const Fr*)S[1], (const Fr*)S[2], (const Fr*)S[4], (u64)H4loc); }
    CTRY(ntt_dist_device(c, S[0], H4loc, S[1], lg4H, 1));                                      // rhs, cyclic slice
    const uint64_t mask_loc = slice_c(c, S[2], pk.mask.fr(), mask_len);
    PTRY(lincomb(c, S[2], H4loc, {{S[2], mask_loc, HFr::one()}, {S[1], H4loc, HFr::one()}}));  // q_1 slice
    PTRY(div_vanishing(c, S[6], S[2], H4loc, Hloc, S[3]));                                     // h_1 slice (3H / G)
    PTRY(lincomb(c, S[4], Hloc, {{S[2], Hloc, HFr::one()}, {S[6], Hloc, HFr::one()}}));        // slice of the remainder x g_1
    const Fr* rcv = nullptr;
    CTRY(allgather2(c, S[6], 2 * Hloc, S[4], Hloc, &rcv));
    { ProfScope ps(c, PF_GLUE);
      unslice_c(c, pk.h1.fr(), rcv, h1_len, 3 * Hloc, 0);
      unslice_c(c, S[4], rcv, H, 3 * Hloc, 2 * Hloc); }
    PHIP(hipGetLastError());
  } else {
  PTRY(ntt_device_len(c, S[5], H, S[0], lg4H, 0));                                     // r_alpha
  // summed_z_m: already in S[1]
  if (!side_ntt) PTRY(ntt_device_len(c, S[7], z_len, S[2], lg4H, 0));                  // z (else: already there, beside round 1's commitment)
  PTRY(ntt_device_len(c, pk.t.fr(), H, S[4], lg4H, 0));                                // t
  { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::mul_sub_mul_kernel, H4, S[0], (const Fr*)S[0], (const Fr*)S[1], (const Fr*)S[2], (const Fr*)S[4], (u64)H4); }
  PTRY(ntt_device(c, S[0], S[1], lg4H, 1));                                  // rhs
  PTRY(lincomb(c, S[2], H4, {{pk.mask.fr(), mask_len, HFr::one()}, {S[1], H4, HFr::one()}}));      // q_1 = mask + rhs
  // (h_1, x g_1) = q_1 / v_H (prover.rs:550-551)
  PTRY(div_vanishing(c, pk.h1.fr(), S[2], H4, H, S[3]));
  PTRY(lincomb(c, S[4], H, {{S[2], H, HFr::one()}, {pk.h1.fr(), H, HFr::one()}}));   // remainder = x g_1
  }
  PTRY(d2d(c, pk.g1.fr(), S[4] + 1, H - 1));
  const uint64_t g1_len = H - 1;
  tr.mark("AHP::Prover::SecondRound"); mh::host_tick("round 2 kernels issued");
  std::vector<fsh::Commitment> cm2; std::vector<PolyRand> rd2;
  CTRY(marlin_commit(c, pk, {{pk.t.fr(), H, false, 0, false}, {pk.g1.fr(), g1_len, true, H - 2, true},
                               {pk.h1.fr(), h1_len, false, 0, false}}, &zk, cm2, rd2));
  fsh::Commitment &c_t = cm2[0], &c_g1 = cm2[1], &c_h1 = cm2[2];
  PolyRand& rd_g1 = rd2[1];
  {
    std::vector<uint8_t> b;
    put_comm(b, c_t, pk.pc); put_comm(b, c_g1, pk.pc); put_comm(b, c_h1, pk.pc);
    fs.absorb(b);                                                               // lib.rs:201
  }
  tr.mark("Committing to second round polys"); mh::host_tick("round 2 committed + absorbed");
  HFr beta = fs.rand_fr();
  while (v_h(beta).is_zero()) beta = fs.rand_fr();                              // verifier.rs:82-91

  // ---------------- third round (prover.rs:588-706) ----------------------------------------------------------
  HFr vH_beta = v_h(beta);
  HFr vv = vH_alpha * vH_beta;
  HFr ea = eta_a * vv, eb = eta_b * vv, ec = eta_c * vv;
  // The reference interpolates b (655) and f (681) from their evaluations on K, multiplies them through three
  // transforms of size 2K (685) and divides a - b f by v_K (686).  The quotient h_2 is the same polynomial whichever way
  // it is computed, and a - b f vanishes on K by construction of f (f(k) = a(k) / b(k)), so here it is evaluated on the
  // coset g K, where v_K is the constant g^K - 1:  a = sum_M e_M val_M and b = alpha beta - alpha row - beta col + row_col
  // are combinations of index polynomials whose coset evaluations are part of the key, f needs one forward transform,
  // h_2 one inverse -- three transforms of size K instead of two of size K and three of size 2K.
  // (The reference zips the coefficient vectors of val_a, val_b, val_c when it forms a, prover.rs:625-637, which truncates a
  // to the shortest of them -- e.g. to nothing for an all-zero matrix.  That cannot change h_2: every variant of a has degree
  // < |K|, so it only enters the REMAINDER of the division by v_K, which the reference discards (686-689); the quotient is
  // that of -b f either way.  tests/test_gpu_marlin.py::test_all_zero_matrix_a_poly_zip_quirk pins it.)
  HFr alpha_beta = alpha * beta;
  if (sliced) {
    // the third round on this rank's block of K (index evaluations pre-sliced into M-layout, prepare_sliced) and its cyclic
    // slice of f and h_2; ONE all-gather returns f and h_2 to every rank
    const uint64_t Kloc = K / Gs;
    { ProfScope ps(c, PF_GLUE);
      KLAUNCH(poly::denom_kernel, Kloc, S[1], (const Fr*)pk.sl_ev[0].p, (const Fr*)pk.sl_ev[1].p, arg(alpha), arg(beta), (u64)Kloc); }
    PTRY(batch_inverse(c, S[1], S[3], Kloc, HFr::one(), 0));
    { ProfScope ps(c, PF_GLUE);
      KLAUNCH(poly::f_evals_kernel, Kloc, S[3], (const Fr*)S[1], (const Fr*)pk.sl_ev[2].p, (const Fr*)pk.sl_ev[3].p, (const Fr*)pk.sl_ev[4].p, arg(ea), arg(eb), arg(ec), (u64)Kloc); }
    CTRY(ntt_dist_device(c, S[3], Kloc, S[4], lgK, 1));                        // f, cyclic slice
    { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::twist_kernel, Kloc, S[5], (const Fr*)S[4], (const Fr*)pk.sl_gpow_hi.p, (const Fr*)pk.sl_gpow_lo.p, (u64)Kloc); }
    CTRY(ntt_dist_device(c, S[5], Kloc, S[6], lgK, 0));                        // f on g K, this rank's block
    {
      HFr vk_inv = (pk.coset_g.pow_u64(K) - HFr::one()).inv();
      ProfScope ps(c, PF_GLUE);
      KLAUNCH(poly::h2_coset_kernel, Kloc, S[5], (const Fr*)S[6], (const Fr*)pk.sl_cs[0].p, (const Fr*)pk.sl_cs[1].p, (const Fr*)pk.sl_cs[2].p,
              (const Fr*)pk.sl_cs[3].p, (const Fr*)pk.sl_cs[4].p, (const Fr*)pk.sl_cs[5].p, arg(ea), arg(eb), arg(ec), arg(alpha), arg(beta),
              arg(alpha_beta), arg(vk_inv), (u64)Kloc);
    }
    CTRY(ntt_dist_device(c, S[5], Kloc, S[6], lgK, 1));
    { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::twist_kernel, Kloc, S[7], (const Fr*)S[6], (const Fr*)pk.sl_ginv_hi.p, (const Fr*)pk.sl_ginv_lo.p, (u64)Kloc); }
    const Fr* rcv = nullptr;
    CTRY(allgather2(c, S[4], Kloc, S[7], Kloc, &rcv));
    { ProfScope ps(c, PF_GLUE);
      unslice_c(c, S[4], rcv, K, 2 * Kloc, 0);
      unslice_c(c, pk.h2.fr(), rcv, K, 2 * Kloc, Kloc); }
    PHIP(hipGetLastError());
    PTRY(d2d(c, pk.g2.fr(), S[4] + 1, K - 1));                                // g_2 = f / X
  } else {
  { ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::denom_kernel, K, S[1], (const Fr*)pk.ev_row.p, (const Fr*)pk.ev_col.p, arg(alpha), arg(beta), (u64)K);
  }
  PTRY(batch_inverse(c, S[1], S[3], K, HFr::one(), 0));
  { ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::f_evals_kernel, K, S[3], (const Fr*)S[1], (const Fr*)pk.ev_val_a.p, (const Fr*)pk.ev_val_b.p, (const Fr*)pk.ev_val_c.p, arg(ea), arg(eb), arg(ec), (u64)K); }
  PTRY(ntt_device(c, S[3], S[4], lgK, 1));                                    // f
  PTRY(d2d(c, pk.g2.fr(), S[4] + 1, K - 1));                                  // g_2 = f / X
  { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::twist_kernel, K, S[5], (const Fr*)S[4], (const Fr*)pk.gpow_hi.p, (const Fr*)pk.gpow_lo.p, (u64)K); }
  PTRY(ntt_device(c, S[5], S[6], lgK, 0));                                    // f on g K
  {
    HFr vk_inv = (pk.coset_g.pow_u64(K) - HFr::one()).inv();
    ProfScope ps(c, PF_GLUE);
    KLAUNCH(poly::h2_coset_kernel, K, S[5], (const Fr*)S[6], (const Fr*)pk.cs_val_a.p, (const Fr*)pk.cs_val_b.p, (const Fr*)pk.cs_val_c.p,
            (const Fr*)pk.cs_row.p, (const Fr*)pk.cs_col.p, (const Fr*)pk.cs_row_col.p, arg(ea), arg(eb), arg(ec), arg(alpha), arg(beta),
            arg(alpha_beta), arg(vk_inv), (u64)K);
  }
  PTRY(ntt_device(c, S[5], S[6], lgK, 1));
  { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::twist_kernel, K, pk.h2.fr(), (const Fr*)S[6], (const Fr*)pk.ginv_hi.p, (const Fr*)pk.ginv_lo.p, (u64)K); }
  }
  const uint64_t g2_len = K - 1;
  const uint64_t h2_len = K - 1;
  tr.mark("AHP::Prover::ThirdRound"); mh::host_tick("round 3 kernels issued");
  std::vector<fsh::Commitment> cm3; std::vector<PolyRand> rd3;
  CTRY(marlin_commit(c, pk, {{pk.g2.fr(), g2_len, true, K - 2, false}, {pk.h2.fr(), h2_len, false, 0, false}}, &zk, cm3, rd3));
  fsh::Commitment &c_g2 = cm3[0], &c_h2 = cm3[1];
  {
    std::vector<uint8_t> b;
    put_comm(b, c_g2, pk.pc); put_comm(b, c_h2, pk.pc);
    fs.absorb(b);                                                                // lib.rs:221
  }
  tr.mark("Committing to third round polys"); mh::host_tick("round 3 committed + absorbed");
  HFr gamma = fs.rand_fr();                                                      // verifier.rs:94-100

  // ---------------- evaluations (lib.rs:272-287): g_1, g_2, t, z_b in label order ---------------------------
  HFr g1_beta, g2_gamma, t_beta, zb_beta;
  {
    // four evaluations, ONE copy back
    Fr* slots = (Fr*)pk.scal.p;
    PTRY(eval_poly(c, pk, pk.g1.fr(), g1_len, beta, nullptr, slots + 0));
    PTRY(eval_poly(c, pk, pk.g2.fr(), g2_len, gamma, nullptr, slots + 1));
    PTRY(eval_poly(c, pk, pk.t.fr(), H, beta, nullptr, slots + 2));
    PTRY(eval_poly(c, pk, pk.zb.fr(), za_len, beta, nullptr, slots + 3));
    uint64_t hv[16];
    PHIP(hipMemcpyAsync(hv, slots, 4 * 32, hipMemcpyDeviceToHost, c.stream));
    PHIP(hipStreamSynchronize(c.stream));
    memcpy(g1_beta.v, hv, 32); memcpy(g2_gamma.v, hv + 4, 32); memcpy(t_beta.v, hv + 8, 32); memcpy(zb_beta.v, hv + 12, 32);
  }
  {
    std::vector<uint8_t> b;
    fsh::put_fr(b, g1_beta); fsh::put_fr(b, g2_gamma); fsh::put_fr(b, t_beta); fsh::put_fr(b, zb_beta);
    fs.absorb(b);                                                                // lib.rs:289
  }
  HFr xi = fs.rand_u128_as_fr();                                                 // lib.rs:290

  // ---------------- construct_linear_combinations (mod.rs:110-221) --------------------------------------------
  // r_alpha(beta) = (v_H(alpha) - v_H(beta)) / (alpha - beta)
  HFr r_ab = (alpha == beta) ? HFr::from_u64(H) * alpha.pow_u64(H - 1) : (vH_alpha - vH_beta) * (alpha - beta).inv();
  HFr vX_beta = beta.pow_u64(X) - HFr::one();
  HFr vK_gamma = gamma.pow_u64(K) - HFr::one();
  HFr c_za_lc = r_ab * (eta_a + eta_c * zb_beta);
  HFr c_w_lc = (t_beta * vX_beta).neg();
  HFr c_h1_lc = vH_beta.neg();
  HFr mult = gamma * g2_gamma + t_beta * HFr::from_u64(K).inv();
  // sliced openings (MarlinKZG10, the usual SRS geometry): every rank builds only ITS block of the linear combinations and of
  // the witness polynomials (see below), so the full outer / inner polynomials are never formed
  static const bool sliced_open_env = [] { const char* e = getenv("MH_SLICED_OPEN"); return !e || atoi(e) != 0; }();
  const bool sliced_open = sliced && sliced_open_env && pk.pc == 0 && pk.srs_max_degree - (K - 2) == 1 &&
                           std::max(mask_len - 1, pk.srs_max_degree - (H - 2) + (H - 2)) <= pk.S[1].bytes / 32;
  if (!sliced_open) {
  PTRY(lincomb(c, pk.outer.fr(), mask_len, {{pk.mask.fr(), mask_len, HFr::one()}, {pk.za.fr(), za_len, c_za_lc},
                                               {pk.w.fr(), w_len, c_w_lc}, {pk.h1.fr(), h1_len, c_h1_lc}}));
  PTRY(lincomb(c, pk.inner.fr(), K, {{pk.p_a_val.fr(), K, ea}, {pk.p_b_val.fr(), K, eb}, {pk.p_c_val.fr(), K, ec},
                                       {pk.p_row.fr(), K, alpha * mult}, {pk.p_col.fr(), K, beta * mult}, {pk.p_row_col.fr(), K, mult.neg()},
                                       {pk.h2.fr(), h2_len, vK_gamma.neg()}}));
  }

  tr.mark("Evaluating linear combinations over query set"); mh::host_tick("evaluations + linear combinations issued");
  // ---------------- PC::open_combinations (lib.rs:292) -> batch_open -> MarlinKZG10::open ---------------------
  auto xi_pow = [&](unsigned e) { return xi.pow_u64(e); };
  auto sg = c.bases.find(pk.srs_g);
  if (sg == c.bases.end()) return fail(MH_EINVAL, "prover key refers to a freed SRS handle");
  const char* srs_pts = (const char*)sg->second.d_points;
  HG1Affine w_beta, w_gamma; bool has_rv_beta = false, has_rv_gamma = false; HFr rv_beta = HFr::zero();
  if (pk.pc == 1) {
    // SonicKZG10::open [SURVEY B-5]: per point ONE combined polynomial sum_i xi^i p_i (labels in BTreeSet order) and one
    // KZG10::open on the unshifted powers.  beta: g_1, outer_sumcheck, t, z_b; gamma: g_2, inner_sumcheck.
    PTRY(lincomb(c, S[0], mask_len, {{pk.g1.fr(), g1_len, HFr::one()}, {pk.outer.fr(), mask_len, xi_pow(1)},
                                       {pk.t.fr(), H, xi_pow(2)}, {pk.zb.fr(), za_len, xi_pow(3)}}));
    PTRY(div_linear(c, S[1], S[0], mask_len, beta, S[2]));
    PTRY(lincomb(c, S[0], K, {{pk.g2.fr(), g2_len, HFr::one()}, {pk.inner.fr(), K, xi_pow(1)}}));
    PTRY(div_linear(c, S[5], S[0], K, gamma, S[2]));
    // the hiding part of the witness at beta -- three points of gamma_g against the quotient of the combined blinding polynomial --
    // is multiplied on a host thread WHILE the device runs the batch (through round 6's first half it ran after the batch: 0.2 ms
    // with the GPU idle, 2.5 % of a 2^16 proof)
    std::vector<HFr> r;
    host_axpy(r, HFr::one(), rd_g1.rand.blind);
    std::vector<HFr> r_outer; host_axpy(r_outer, c_za_lc, rd_za.rand.blind); host_axpy(r_outer, c_w_lc, rd_w.rand.blind);
    host_axpy(r, xi_pow(1), r_outer);
    host_axpy(r, xi_pow(3), rd_zb.rand.blind);
    const bool r_nonzero = !host_is_zero(r);
    const std::vector<HFr> rw = r_nonzero ? host_div_linear(r, beta) : std::vector<HFr>();
    const HG1Affine* gp_open = pk.gamma_g;
    std::future<HG1> f_rw;
    if (r_nonzero) f_rw = host_pool().submit([gp_open, &rw] { return small_msm(gp_open, rw); });
    WaitAll wait_open{&f_rw};               // `rw` lives in this frame: no return below may leave a worker reading it
    std::vector<HG1> om;
    CTRY(sharded_msm_batch(c, {{srs_pts, S[1], mask_len - 1}, {srs_pts, S[5], K - 1}}, om));
    HG1 wacc = om[0];
    if (r_nonzero) {
      wacc = wacc.add(f_rw.get());
      rv_beta = host_eval(r, beta); has_rv_beta = true;
    }
    { const HG1 two[2] = {wacc, om[1]}; HG1Affine a2[2]; hostff::batch_to_affine(two, 2, a2); w_beta = a2[0]; w_gamma = a2[1]; }
    // nothing hiding at gamma: random_v = None
  } else {
  const uint64_t off_b = pk.srs_max_degree - (H - 2), off_g = pk.srs_max_degree - (K - 2);
  const bool fuse_g = off_g == 1;            // see below
  const uint64_t wb_len = mask_len - 1, swb_len = g1_len - 1, wg_len = K - 1, swg_len = g2_len - 1;
  const uint64_t mb_len = std::max(wb_len, off_b + swb_len), mg_len = std::max(wg_len, off_g + swg_len);
  // randomness: r = xi^0 rand(g_1) + xi^2 (c_za rand(z_a) + c_w rand(w)) + xi^4 rand(z_b); its witness and the shifted
  // one are multiplied on host threads while the device runs the batch -- submitted before EITHER form of the batch below (through
  // round 6's first half the sliced form ran its batch first and then waited ~0.3 ms for these two with the GPU idle)
  std::vector<HFr> r;
  host_axpy(r, HFr::one(), rd_g1.rand.blind);
  {
    std::vector<HFr> r_outer; host_axpy(r_outer, c_za_lc, rd_za.rand.blind); host_axpy(r_outer, c_w_lc, rd_w.rand.blind);
    host_axpy(r, xi_pow(2), r_outer);
  }
  host_axpy(r, xi_pow(4), rd_zb.rand.blind);
  const bool r_nonzero = !host_is_zero(r);
  const std::vector<HFr> rw = r_nonzero ? host_div_linear(r, beta) : std::vector<HFr>();
  std::vector<HFr> srw; host_axpy(srw, xi_pow(1), host_div_linear(rd_g1.shifted.blind, beta));
  const HG1Affine* gp_open = pk.gamma_g;
  std::future<HG1> f_rw, f_srw = host_pool().submit([gp_open, &srw] { return small_msm(gp_open, srw); });
  if (r_nonzero) f_rw = host_pool().submit([gp_open, &rw] { return small_msm(gp_open, rw); });
  WaitAll wait_open{&f_rw, &f_srw};       // `rw` and `srw` live in this frame: no return below may leave a worker reading them
  std::vector<HG1> om;                       // [0] witness at beta (+ shifted), [1] witness at gamma (+ shifted)
  if (sliced_open) {
    // Point-sharded openings.  The two merged witness vectors (index spaces [0, mb_len) and [0, mg_len) of the SRS) are cut
    // into G contiguous blocks; rank r builds block r of the combined polynomial straight from the replicated round
    // polynomials (one lincomb over 1 / G of the range: the 7-term inner combination over K is the dearest glue kernel of a
    // proof), divides it by (X - z) locally -- the recurrence q_i = p_{i+1} + z q_{i+1} only needs q at the block's upper
    // end, which is sum_{r' > r} E_{r'} z^(a_{r'} - a_{r+1}) with E_{r'} = the value of block r' at z: ONE all_gather of two
    // field elements per rank -- and multiplies its block against the matching base range; the partial points are combined
    // like bucket-range shares.  The shifted witness of g_1 (H - 2 coefficients against shifted_powers) is divided the same
    // way, its carry taken directly from the replicated g_1.
    const uint64_t G = Gs, r = (uint64_t)g_shard.rank;
    auto cut = [&](uint64_t L, uint64_t cap, uint64_t k) { return std::min<uint64_t>(L * k / G, cap); };
    auto clip = [](const Fr* p, uint64_t len, uint64_t a, uint64_t e, const HFr& coef) { return Term{p + a, len > a ? std::min(len, e) - a : 0, coef}; };
    // boundaries in the index space of the divided polynomial (length len, quotient length len - 1): a_k, a_G = len
    auto bounds = [&](uint64_t L, uint64_t qlen, uint64_t len, uint64_t k) { return k >= G ? len : cut(L, qlen, k); };
    const uint64_t ab = bounds(mb_len, wb_len, mask_len, r), eb_ = bounds(mb_len, wb_len, mask_len, r + 1);     // beta: block [ab, eb_)
    const uint64_t ag = bounds(mg_len, wg_len, K, r), eg = bounds(mg_len, wg_len, K, r + 1);                    // gamma
    // blocks of the combined polynomials: beta in S[0], gamma in S[3]
    PTRY(lincomb(c, S[0], eb_ - ab, {clip(pk.g1.fr(), g1_len, ab, eb_, HFr::one()), clip(pk.mask.fr(), mask_len, ab, eb_, xi_pow(2)),
                                       clip(pk.za.fr(), za_len, ab, eb_, xi_pow(2) * c_za_lc), clip(pk.w.fr(), w_len, ab, eb_, xi_pow(2) * c_w_lc),
                                       clip(pk.h1.fr(), h1_len, ab, eb_, xi_pow(2) * c_h1_lc), clip(pk.t.fr(), H, ab, eb_, xi_pow(3)),
                                       clip(pk.zb.fr(), za_len, ab, eb_, xi_pow(4))}));
    const HFr x2 = xi_pow(2);
    PTRY(lincomb(c, S[3], eg - ag, {clip(pk.g2.fr(), g2_len, ag, eg, HFr::one()), clip(pk.p_a_val.fr(), K, ag, eg, x2 * ea),
                                      clip(pk.p_b_val.fr(), K, ag, eg, x2 * eb), clip(pk.p_c_val.fr(), K, ag, eg, x2 * ec),
                                      clip(pk.p_row.fr(), K, ag, eg, x2 * alpha * mult), clip(pk.p_col.fr(), K, ag, eg, x2 * beta * mult),
                                      clip(pk.p_row_col.fr(), K, ag, eg, x2 * mult.neg()), clip(pk.h2.fr(), h2_len, ag, eg, x2 * vK_gamma.neg())}));
    // + xi X g_2: F_i += xi g_2[i - 1] for 1 <= i <= g2_len, on this block
    {
      const uint64_t lo = std::max<uint64_t>(ag, 1), hi = std::min<uint64_t>(eg, g2_len + 1);
      if (hi > lo) { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::axpy_shifted_kernel, hi - lo, S[3], (const Fr*)(pk.g2.fr() + (lo - 1)), arg(xi_pow(1)), (u64)(lo - ag), (u64)(hi - lo)); }
    }
    // block values at the points, exchanged
    HFr Eb = HFr::zero(), Eg = HFr::zero();
    if (eb_ > ab) PTRY(eval_poly(c, pk, S[0], eb_ - ab, beta, &Eb));
    if (eg > ag) PTRY(eval_poly(c, pk, S[3], eg - ag, gamma, &Eg));
    std::vector<uint64_t> snd(8), all_e(8 * G);
    memcpy(&snd[0], Eb.v, 32); memcpy(&snd[4], Eg.v, 32);
    {
      ExchangeScope xs(c);
      if (g_shard.cb(snd.data(), 64, all_e.data(), g_shard.user) != 0) { g_job.failed = true; return fail(MH_EHIP, "sliced openings: all_gather failed: " + g_err); }
    }
    auto carry = [&](int which, const HFr& z, uint64_t L, uint64_t qlen, uint64_t len) {
      HFr acc = HFr::zero();
      const uint64_t top = bounds(L, qlen, len, r + 1);
      for (uint64_t q = r + 1; q < G; q++) {
        HFr e; memcpy(e.v, &all_e[8 * q + 4 * which], 32);
        acc = acc + e * z.pow_u64(bounds(L, qlen, len, q) - top);
      }
      return acc;
    };
    // quotient blocks: the block extended by the carry (the top rank's block already ends with the leading coefficient)
    auto divide_block = [&](Fr* q, Fr* blk, uint64_t a, uint64_t e, uint64_t qlen, const HFr& z, const HFr& car) -> int {
      const uint64_t qb = std::min(e, qlen);                 // quotient indices [a, qb)
      if (qb <= a) return MH_OK;
      uint64_t n = e - a;
      if (r + 1 < G) { PTRY(set_fr(c, blk + (qb - a), car)); n = qb - a + 1; }
      return div_linear(c, q, blk, n, z, S[2]);
    };
    const uint64_t lo_b = mb_len * r / G, hi_b = mb_len * (r + 1) / G;     // this rank's block of the merged beta vector
    PHIP(hipMemsetAsync(S[1], 0, (hi_b - lo_b + 1) * 32, c.stream));
    if (ab >= lo_b) PTRY(divide_block(S[1] + (ab - lo_b), S[0], ab, eb_, wb_len, beta, carry(0, beta, mb_len, wb_len, mask_len)));   // else: no main-witness index in this block
    // shifted witness of g_1 on [lo_b, hi_b) n [off_b, off_b + swb_len): quotient indices [ka, kb) of g_1 / (X - beta)
    {
      const uint64_t ka = std::max(lo_b, off_b) - off_b, kb = std::min(std::max(hi_b, off_b), off_b + swb_len) - off_b;
      if (kb > ka) {
        HFr car = HFr::zero();
        PTRY(eval_poly(c, pk, pk.g1.fr() + kb, g1_len - kb, beta, &car));                 // sum_{j >= kb} g_1[j] beta^(j - kb)
        PTRY(d2d(c, S[4], pk.g1.fr() + ka, kb - ka));
        PTRY(set_fr(c, S[4] + (kb - ka), car));
        PTRY(div_linear(c, S[6], S[4], kb - ka + 1, beta, S[2]));
        ProfScope ps(c, PF_GLUE);
        KLAUNCH(poly::axpy_shifted_kernel, kb - ka, S[1], (const Fr*)S[6], arg(xi_pow(1)), (u64)(off_b + ka - lo_b), (u64)(kb - ka));
      }
    }
    PTRY(divide_block(S[5], S[3], ag, eg, wg_len, gamma, carry(1, gamma, mg_len, wg_len, K)));
    if (r == 0 && g_job.poison == MH_OK) hipLaunchKernelGGL(poly::add_const_kernel, dim3(1), dim3(64), 0, c.stream, S[5], arg((xi_pow(1) * g2_gamma).neg()));
    PHIP(hipGetLastError());
    const uint64_t qg_end = std::min(eg, wg_len);
    std::vector<MsmJob> jobs;
    jobs.push_back({srs_pts + lo_b * PT_B, S[1], hi_b - lo_b});
    jobs.push_back({srs_pts + ag * PT_B, S[5], qg_end > ag ? qg_end - ag : 0});
    std::vector<HG1> res;
    CTRY(sharded_msm_batch(c, jobs, res, 1, true));
    om = {res[0], res[1]};
  }
  // The MSMs of the two opening proofs (witness + shifted witness at beta and at gamma, merged below) run as one batch.
  // --- at beta: labels g_1, outer_sumcheck, t, z_b  -> challenges xi^0 (g_1), xi^1 (g_1 shifted), xi^2, xi^3, xi^4
  if (!sliced_open) {
  PTRY(lincomb(c, S[0], mask_len, {{pk.g1.fr(), g1_len, HFr::one()}, {pk.outer.fr(), mask_len, xi_pow(2)},
                                     {pk.t.fr(), H, xi_pow(3)}, {pk.zb.fr(), za_len, xi_pow(4)}}));
  PTRY(div_linear(c, S[1], S[0], mask_len, beta, S[2]));                         // witness of the combined polynomial
  PTRY(div_linear(c, S[3], pk.g1.fr(), g1_len, beta, S[2]));                     // degree-bounded g_1: shifted witness
  PTRY(lincomb(c, S[4], g1_len - 1, {{S[3], g1_len - 1, xi_pow(1)}}));
  // --- at gamma: labels g_2, inner_sumcheck -> challenges xi^0 (g_2), xi^1 (g_2 shifted), xi^2
  // When shifted_powers(K - 2) starts at SRS index 1 (max_degree = K - 1, the usual case) the witness plus the shifted
  // witness moved up by one coefficient is the quotient of ONE polynomial: with C = g_2 + xi^2 inner,
  //   (C - C(gamma)) / (X - gamma) + X xi (g_2 - g_2(gamma)) / (X - gamma) = quotient(C + xi X g_2) - xi g_2(gamma),
  // so a single division serves both (the general offset keeps two divisions and adds the vectors).
  PTRY(lincomb(c, S[0], K, {{pk.g2.fr(), g2_len, HFr::one()}, {pk.inner.fr(), K, xi_pow(2)}}));
  if (fuse_g) {
    { ProfScope ps(c, PF_GLUE); KLAUNCH(poly::axpy_shifted_kernel, g2_len, S[0], (const Fr*)pk.g2.fr(), arg(xi_pow(1)), (u64)1, (u64)g2_len); }
    PTRY(div_linear(c, S[5], S[0], K, gamma, S[2]));
    if (g_job.poison == MH_OK) hipLaunchKernelGGL(poly::add_const_kernel, dim3(1), dim3(64), 0, c.stream, S[5], arg((xi_pow(1) * g2_gamma).neg()));
  } else {
    PTRY(div_linear(c, S[5], S[0], K, gamma, S[2]));
    PTRY(div_linear(c, S[3], pk.g2.fr(), g2_len, gamma, S[2]));
    PTRY(lincomb(c, S[6], g2_len - 1, {{S[3], g2_len - 1, xi_pow(1)}}));
  }
  }
  // The reference multiplies witness and shifted witness separately (kzg10::open on powers and on shifted_powers) and
  // adds the two commitments (marlin_pc open: w = w + shifted_w).  shifted_powers(d) is the same SRS array from index
  // max_degree - d, so the sum is ONE multi-scalar multiplication with the shifted witness's coefficients added at
  // that offset: at gamma the offset is max_degree - (K - 2) (1 for this circuit: the two 4M-point MSMs collapse into
  // one), at beta the ranges do not overlap but one job replaces two.
  // (an SRS much larger than this index needs puts the shifted range beyond the scratch vectors: then the two stay apart)
  const bool merge_b = mb_len <= pk.S[1].bytes / 32, merge_g = fuse_g || mg_len <= pk.S[5].bytes / 32;
  if (!sliced_open) {
  if (merge_b) PTRY(zero_tail(c, S[1], wb_len, mb_len));
  if (merge_g && !fuse_g) PTRY(zero_tail(c, S[5], wg_len, mg_len));
  { ProfScope ps(c, PF_GLUE);
    if (merge_b) KLAUNCH(poly::add_shifted_kernel, swb_len, S[1], (const Fr*)S[4], (u64)off_b, (u64)swb_len);
    if (merge_g && !fuse_g) KLAUNCH(poly::add_shifted_kernel, swg_len, S[5], (const Fr*)S[6], (u64)off_g, (u64)swg_len); }
  {
    std::vector<MsmJob> jobs;
    jobs.push_back({srs_pts, S[1], merge_b ? mb_len : wb_len});
    jobs.push_back({srs_pts, S[5], merge_g ? mg_len : wg_len});
    if (!merge_b) jobs.push_back({srs_pts + off_b * PT_B, S[4], swb_len});
    if (!merge_g) jobs.push_back({srs_pts + off_g * PT_B, S[6], swg_len});
    std::vector<HG1> res;
    CTRY(sharded_msm_batch(c, jobs, res));
    om = {res[0], res[1]};
    size_t nx = 2;
    if (!merge_b) om[0] = om[0].add(res[nx++]);
    if (!merge_g) om[1] = om[1].add(res[nx++]);
  }
  }
  HG1 w_beta_jac;
  {
    HG1 wacc = om[0];
    if (r_nonzero) {
      wacc = wacc.add(f_rw.get());
      rv_beta = host_eval(r, beta); has_rv_beta = true;
    }
    HG1 sw = HG1::identity();     // the device part of the shifted witness is inside om[0]
    // open_with_witness_polynomial(shifted_powers, point, shifted_r, shifted_w, Some(shifted_r_witness)):
    // the hiding witness is always Some(..) on this path, so random_v is always Some(shifted_r(point))
    std::vector<HFr> sr; host_axpy(sr, xi_pow(1), rd_g1.shifted.blind);
    {
      sw = sw.add(f_srw.get());
      // marlin_pc `open`: `if let Some(s) = shifted_proof.random_v { random_v = random_v.map(|v| v + s) }` -- the shifted
      // evaluation is added to a Some and never turns a None into a Some [ark-poly-commit 0.3, SURVEY B-4]
      if (has_rv_beta) rv_beta = rv_beta + host_eval(sr, beta);
    }
    w_beta_jac = wacc.add(sw);
  }
  { const HG1 two[2] = {w_beta_jac, om[1]}; HG1Affine a2[2]; hostff::batch_to_affine(two, 2, a2); w_beta = a2[0]; w_gamma = a2[1]; }
  // at gamma nothing is hiding: kzg10::open on the unshifted powers returns random_v = None, and the Some(0) of the
  // shifted proof (open_with_witness_polynomial with Some(empty witness)) is dropped by `random_v.map(..)`
  // [ark-poly-commit 0.3 marlin_pc::open, SURVEY B-4]
  has_rv_gamma = false;
  }

  tr.mark("PC::open_combinations"); mh::host_tick("openings done"); mh::host_ticks_flush();
  pk.last_polys = {{"w", {pk.w.fr(), w_len}}, {"z_a", {pk.za.fr(), za_len}}, {"z_b", {pk.zb.fr(), za_len}},
                   {"mask_poly", {pk.mask.fr(), mask_len}}, {"t", {pk.t.fr(), H}}, {"g_1", {pk.g1.fr(), g1_len}},
                   {"h_1", {pk.h1.fr(), h1_len}}, {"g_2", {pk.g2.fr(), g2_len}}, {"h_2", {pk.h2.fr(), h2_len}},
                   {"row", {pk.p_row.fr(), K}}, {"col", {pk.p_col.fr(), K}}, {"a_val", {pk.p_a_val.fr(), K}},
                   {"b_val", {pk.p_b_val.fr(), K}}, {"c_val", {pk.p_c_val.fr(), K}}, {"row_col", {pk.p_row_col.fr(), K}}};
  if (g_job.poison != MH_OK) return job_error();   // (unreachable after a completed last all-gather; kept as the last line of defence)
  if (zk.bad) return fail(MH_EINVAL, "mh_marlin_prove_draws: too few zk draws, or a draw that is not a reduced field element");
  // ---------------- Proof (lib.rs:305-310), flat ToBytes layout ------------------------------------------------
  std::vector<uint8_t> out;
  fsh::Commitment* all[9] = {&c_w, &c_za, &c_zb, &c_mask, &c_t, &c_g1, &c_h1, &c_g2, &c_h2};
  for (auto* cm : all) put_comm(out, *cm, pk.pc);
  fsh::put_fr(out, g1_beta); fsh::put_fr(out, g2_gamma); fsh::put_fr(out, t_beta); fsh::put_fr(out, zb_beta);
  fsh::put_g1(out, w_beta);
  out.push_back(has_rv_beta ? 1 : 0); fsh::put_fr(out, has_rv_beta ? rv_beta : HFr::zero());
  fsh::put_g1(out, w_gamma);
  out.push_back(has_rv_gamma ? 1 : 0); fsh::put_fr(out, HFr::zero());
  if (len_out) *len_out = out.size();
  if (cap < out.size()) return fail(MH_EINVAL, "mh_marlin_prove: proof buffer too small");
  memcpy(proof_out, out.data(), out.size());
  return MH_OK;
}
#undef KLAUNCH
#define KLAUNCH(kern, n, ...)                                                                          \
  do {                                                                                                 \
    hipLaunchKernelGGL(kern, dim3(poly::grid_for(n)), dim3(poly::TPB), 0, c.stream, __VA_ARGS__);      \
  } while (0)

}  // extern "C"
