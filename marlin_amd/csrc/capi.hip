// libmarlin_hip.so -- C ABI entry points (include/marlin_hip.h) and the host-side
// drivers that sequence the gfx950 kernels.
#include "drivers.h"
#include "ff.cuh"
#include "g1.cuh"
#include "msm.cuh"
#include "msm_tree.cuh"
#include "msm_fb.cuh"
#include "msm_check.cuh"
#include "msm_g2.cuh"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "ntt.cuh"
#include "ntt30.cuh"
#include "srs.cuh"
#include "host_ff.h"
#include "host_pool.h"

using namespace mh;
using hostff::HFq;
using hostff::HFr;
using hostff::HG1;
using hostff::HG1Affine;
using hostff::FQ_L; using hostff::FQ_B; using hostff::PT_B; using hostff::AFF_L; using hostff::XYZ_L; using hostff::XYZZ_L;

namespace mh {
thread_local std::string g_err;
int g_debug_fail_scratch = 0;
int g_debug_poison_scratch = 0;
int g_debug_corrupt = 0;
uint64_t g_debug_scratch_calls = 0;
bool g_multi_ctx = false;
// handles (base sets, prover keys) are unique across the process, not per context: one that reaches the wrong context (a binding's
// finaliser running on another thread) is unknown there instead of naming somebody else's object
std::atomic<uint64_t> g_next_handle{1};
static thread_local Context* t_ctx = nullptr;
// never destroyed: at process exit the HIP runtime may be gone before this library's statics
Context& default_ctx() { static Context* c = new Context(); return *c; }
Context& ctx() { return t_ctx ? *t_ctx : default_ctx(); }

static Fr to_dev_fr(const HFr& h) {
  Fr r;
  memcpy(r.v, h.v, 32);
  return r;
}

// --------------------------------------------------------------------------------
// NTT driver
// --------------------------------------------------------------------------------
static bool ntt_shoup();
static int ensure_twiddles(Context& c, uint32_t log_n) {
  if (c.tw && c.tw_log >= log_n) return MH_OK;
  uint32_t want = log_n < 16 ? 16 : log_n;
  if (c.tw) { MH_HIP(hipStreamSynchronize(c.stream)); (void)hipFree(c.tw); c.tw = nullptr; c.tw_log = 0; }
  MH_HIP(hipMalloc(&c.tw, (size_t)32 << want));
  uint64_t n = 1ull << want;
  Fr root = to_dev_fr(hostff::fr_two_adic_root());
  hipLaunchKernelGGL(ntt::build_twiddles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, (Fr*)c.tw, want, root);
  MH_HIP(hipGetLastError());
  // the same table as w R' mod r on 30-bit limbs (36 B per entry) for the butterflies of ntt30.cuh
  if (c.tw30) { (void)hipFree(c.tw30); c.tw30 = nullptr; }
  MH_HIP(hipMalloc(&c.tw30, (size_t)36 << want));
  hipLaunchKernelGGL(ntt30::build_twiddles30, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, (u32*)c.tw30, (const Fr*)c.tw, (u64)n);
  MH_HIP(hipGetLastError());
  if (c.tw30s) { (void)hipFree(c.tw30s); c.tw30s = nullptr; }
  if (ntt_shoup()) {
    MH_HIP(hipMalloc(&c.tw30s, (size_t)72 << want));
    hipLaunchKernelGGL(ntt30::build_twiddles30s, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, (u32*)c.tw30s, (const Fr*)c.tw, (u64)n);
    MH_HIP(hipGetLastError());
  }
  c.tw_log = want;
  return MH_OK;
}

int ensure_twiddles_public(Context& c, uint32_t log_n) { return ensure_twiddles(c, log_n); }

static void plan_passes(uint32_t log_n, uint32_t* bits, int* npass) {
  int np = (int)((log_n + ntt::MAX_B - 1) / ntt::MAX_B);
  if (np == 0) np = 1;
  uint32_t base = log_n / np, rem = log_n % np;
  for (int i = 0; i < np; i++) bits[i] = base + ((uint32_t)i < rem ? 1 : 0);
  *npass = np;
}

// MH_NTT=32 selects the 32-bit-limb radix-2 kernel of ntt.cuh (the cross-check of the 30-bit radix-8 kernel of ntt30.cuh).  The
// 30-bit kernel takes its twiddle products as Shoup multiplications (ntt30.cuh: butterfly_shoup) in the one-stage rounds -- round 4
// measured it behind a switch (-0.2 ... -0.35 ms per proof at 2^20, profiles/r04zh_*), round 5 ran the whole suite on it and made it
// the default -- and as Montgomery products in the two-stage rounds of the >= 2^23-point transforms (wider twiddles cost registers there).
static bool ntt_use30() { static const bool v = [] { const char* e = getenv("MH_NTT"); return !(e && atoi(e) == 32); }(); return v; }
static bool ntt_shoup() { return ntt_use30(); }

static int launch_pass(Context& c, const Fr* x, Fr* y, uint32_t log_n, uint32_t B, uint32_t logP, uint32_t flags,
                       const Fr& ninv, uint64_t in_len) {
  uint32_t logc = log_n - B;
  if (logc > (uint32_t)ntt::MAX_LOGC) logc = ntt::MAX_LOGC;
  if (ntt_use30()) {
    const uint64_t blocks30 = (1ull << (log_n - B)) >> logc;
    const u32* tw30 = (const u32*)c.tw30;
    // stages per register round: 2 from 2^23 points on, 1 below (measured, see ntt30.cuh)
    const int ns = log_n >= 23 ? 2 : 1;
#define NTT30_LAUNCH_T(LC, NS, SH, TW) hipLaunchKernelGGL((ntt30::pass30_kernel<LC, NS, 256, 4, 0, false, SH>), dim3((unsigned)blocks30), dim3(256), \
    ntt30::pass_lds_bytes(B, (int)logc, logP == 0, 0, false), c.stream, x, y, TW, log_n, B, logP, flags, ninv, (u64)in_len)
#define NTT30_LAUNCH(LC, NS) NTT30_LAUNCH_T(LC, NS, false, tw30)
#define NTT30_LAUNCH_S(LC) do { if (shoup) NTT30_LAUNCH_T(LC, 1, true, tw30s); else NTT30_LAUNCH_T(LC, 1, false, tw30); } while (0)
    // Shoup twiddle products only in the one-stage rounds: with two stages per round the 18-word twiddles cost more registers
    // than the shorter products give back (2^22 constraints: 15.2 -> 16.1 ms of transforms, profiles/r04zj_*)
    const bool shoup = ntt_shoup() && ns == 1;
    const u32* tw30s = (const u32*)c.tw30s;
    if (ns == 2) { switch (logc) { case 0: NTT30_LAUNCH(0, 2); break; case 1: NTT30_LAUNCH(1, 2); break; default: NTT30_LAUNCH(2, 2); break; } }
    else { switch (logc) { case 0: NTT30_LAUNCH_S(0); break; case 1: NTT30_LAUNCH_S(1); break; default: NTT30_LAUNCH_S(2); break; } }
    MH_HIP(hipGetLastError());
    return MH_OK;
  }
  uint64_t blocks = (1ull << (log_n - B)) >> logc;
  size_t lds = ntt::pass_lds_bytes(B, (int)logc);
  dim3 grid((unsigned)blocks), block(ntt::THREADS);
  switch (logc) {
    case 0: hipLaunchKernelGGL(ntt::pass_kernel<0>, grid, block, lds, c.stream, x, y, (const Fr*)c.tw, log_n, B, logP, flags, ninv, (u64)in_len); break;
    case 1: hipLaunchKernelGGL(ntt::pass_kernel<1>, grid, block, lds, c.stream, x, y, (const Fr*)c.tw, log_n, B, logP, flags, ninv, (u64)in_len); break;
    case 2: hipLaunchKernelGGL(ntt::pass_kernel<2>, grid, block, lds, c.stream, x, y, (const Fr*)c.tw, log_n, B, logP, flags, ninv, (u64)in_len); break;
    default: hipLaunchKernelGGL(ntt::pass_kernel<3>, grid, block, lds, c.stream, x, y, (const Fr*)c.tw, log_n, B, logP, flags, ninv, (u64)in_len); break;
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

static int ntt_set_attrs(Context& c) {
  if (c.ntt_attr_done) return MH_OK;
  int lds = (int)ntt::pass_lds_bytes(ntt::MAX_B, ntt::MAX_LOGC);
  MH_HIP(hipFuncSetAttribute((const void*)ntt::pass_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  MH_HIP(hipFuncSetAttribute((const void*)ntt::pass_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  MH_HIP(hipFuncSetAttribute((const void*)ntt::pass_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  MH_HIP(hipFuncSetAttribute((const void*)ntt::pass_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int lds30 = (int)ntt30::pass_lds_bytes(ntt30::MAX_B, ntt30::MAX_LOGC, true, 0, false);
#define NTT30_ATTR(NS) \
  MH_HIP(hipFuncSetAttribute((const void*)ntt30::pass30_kernel<0, NS, 256, 4, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds30)); \
  MH_HIP(hipFuncSetAttribute((const void*)ntt30::pass30_kernel<1, NS, 256, 4, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds30)); \
  MH_HIP(hipFuncSetAttribute((const void*)ntt30::pass30_kernel<2, NS, 256, 4, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds30));
  NTT30_ATTR(1) NTT30_ATTR(2)
  MH_HIP(hipFuncSetAttribute((const void*)ntt30::pass30_kernel<0, 1, 256, 4, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds30));
  MH_HIP(hipFuncSetAttribute((const void*)ntt30::pass30_kernel<1, 1, 256, 4, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds30));
  MH_HIP(hipFuncSetAttribute((const void*)ntt30::pass30_kernel<2, 1, 256, 4, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds30));
  c.ntt_attr_done = true;
  return MH_OK;
}

int ntt_device(Context& c, const void* d_in, void* d_out, uint32_t log_n, int inverse) {
  return ntt_device_len(c, d_in, 1ull << log_n, d_out, log_n, inverse);
}

// d_in may equal d_out.  d_in holds in_len <= 2^log_n elements, the rest of the domain is zero.
int ntt_device_len(Context& c, const void* d_in, uint64_t in_len, void* d_out, uint32_t log_n, int inverse) {
  if (log_n > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "log_n exceeds the two-adicity of Fr");
  size_t bytes = (size_t)32 << log_n;
  if (in_len > (1ull << log_n)) return fail(MH_EINVAL, "ntt: input longer than the domain");
  if (log_n == 0) {
    if (in_len == 0) MH_HIP(hipMemsetAsync(d_out, 0, 32, c.stream));
    else if (d_in != d_out) MH_HIP(hipMemcpyAsync(d_out, d_in, 32, hipMemcpyDeviceToDevice, c.stream));
    return MH_OK;
  }
  MH_TRY(ensure_twiddles(c, log_n));
  MH_TRY(ntt_set_attrs(c));
  uint32_t bits[8];
  int np;
  plan_passes(log_n, bits, &np);
  // n^-1 in Montgomery form
  HFr ninv_h = HFr::from_u64(1ull << (log_n < 63 ? log_n : 0)).inv();
  Fr ninv = to_dev_fr(ninv_h);
  // buffer chain: in -> ... -> out, intermediates in scratch (never writes d_in unless d_in == d_out)
  const Fr* src = (const Fr*)d_in;
  const size_t mid_bytes = ntt_use30() ? (size_t)36 << log_n : bytes;     // intermediate passes of the 30-bit kernel: 36 B per element
  MH_TRY(c.ntt_tmp[0].ensure(np > 1 ? mid_bytes : 0));
  MH_TRY(c.ntt_tmp[1].ensure(np > 2 ? mid_bytes : 0));
  uint32_t logP = 0;
  ProfScope ps(c, PF_NTT);
  for (int p = 0; p < np; p++) {
    Fr* dst;
    if (p == np - 1) dst = (Fr*)d_out;
    else dst = (Fr*)c.ntt_tmp[p & 1].ptr;
    // flags: bit 0 = last pass of an inverse transform (n^-1, index negation), bit 1 = last pass, bit 2 / bit 3 = the
    // pass reads / writes the 36-byte lazy 9-limb elements that the 30-bit kernel keeps between passes
    const bool lazy_mid = ntt_use30();
    const uint32_t pass_flags = ((inverse && p == np - 1) ? 1u : 0u) | (p == np - 1 ? 2u : 0u) | ((lazy_mid && p > 0) ? 4u : 0u) |
                                ((lazy_mid && p < np - 1) ? 8u : 0u);
    // last pass writes d_out; if d_out == src of this pass (only when np == 1 and in-place) bounce via scratch
    if (p == np - 1 && (const void*)dst == (const void*)src) {
      MH_TRY(c.ntt_tmp[0].ensure(bytes));
      dst = (Fr*)c.ntt_tmp[0].ptr;
      MH_TRY(launch_pass(c, src, dst, log_n, bits[p], logP, pass_flags, ninv, p == 0 ? in_len : (1ull << log_n)));
      MH_HIP(hipMemcpyAsync(d_out, dst, bytes, hipMemcpyDeviceToDevice, c.stream));
    } else {
      MH_TRY(launch_pass(c, src, dst, log_n, bits[p], logP, pass_flags, ninv, p == 0 ? in_len : (1ull << log_n)));
    }
    src = dst;
    logP += bits[p];
  }
  return MH_OK;
}

// Radix2EvaluationDomain::{coset_fft_in_place, coset_ifft_in_place} [ark-poly 0.3]: forward = distribute_powers(g) then
// the transform; inverse = the inverse transform then distribute_powers(g^-1); g = F::multiplicative_generator().
int ntt_coset_device(Context& c, const void* d_in, void* d_out, uint32_t log_n, int inverse) {
  const HFr g = HFr::from_u64(hostff::FR_MULT_GENERATOR);
  const uint64_t n = 1ull << log_n;
  const unsigned grid = (unsigned)((n + (uint64_t)256 * ntt::DP_CH - 1) / ((uint64_t)256 * ntt::DP_CH));
  if (!inverse) {
    { ProfScope ps(c, PF_GLUE);
      hipLaunchKernelGGL(ntt::distribute_powers_kernel, dim3(grid), dim3(256), 0, c.stream, (Fr*)d_out, (const Fr*)d_in, to_dev_fr(g), (u64)n); }
    MH_HIP(hipGetLastError());
    return ntt_device(c, d_out, d_out, log_n, 0);
  }
  MH_TRY(ntt_device(c, d_in, d_out, log_n, 1));
  { ProfScope ps(c, PF_GLUE);
    hipLaunchKernelGGL(ntt::distribute_powers_kernel, dim3(grid), dim3(256), 0, c.stream, (Fr*)d_out, (const Fr*)d_out, to_dev_fr(g.inv()), (u64)n); }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// every uploaded base must satisfy the curve equation (an affine point has no infinity flag on this ABI, so an identity
// encoded as (0, 0) or (0, 1) is rejected here instead of being multiplied as if it were a finite point)
__global__ __launch_bounds__(256) void bases_check_kernel(const G1Affine* __restrict__ pts, u64 n, u32* __restrict__ bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const G1Affine p = g1_load_affine(pts + i);
  const Fq rhs = ff_add(ff_mul(ff_sqr(p.x), p.x), G1_CURVE_B_MONT());
  const Fq lhs = ff_sqr(p.y);
  bool same = true;
  for (int k = 0; k < Fq::N; k++) same = same && lhs.v[k] == rhs.v[k];
  if (!same) atomicAdd(bad, 1u);
}
// dst[i] = src[first + i * stride]: the bases of a strided slice as a contiguous vector (the skew fallback of msm_batch_strided_device)
__global__ __launch_bounds__(256) void gather_strided_kernel(G1Affine* __restrict__ dst, const G1Affine* __restrict__ src, u64 first, u64 stride, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const G1Affine p = g1_load_affine(src + first + i * stride);
  ff_store(&dst[i].x, p.x); ff_store(&dst[i].y, p.y);
}
static int bases_validate(Context& c, const void* d_points, size_t n) {
  if (n == 0) return MH_OK;
  MH_TRY(c.tr_sums.ensure(64));
  u32* d_bad = (u32*)c.tr_sums.ptr;
  MH_HIP(hipMemsetAsync(d_bad, 0, 4, c.stream));
  hipLaunchKernelGGL(bases_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, (const G1Affine*)d_points, (u64)n, d_bad);
  MH_HIP(hipGetLastError());
  u32 bad = 0;
  MH_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  if (bad) return fail(MH_EINVAL, "bases: " + std::to_string(bad) + " point(s) are not on the curve (the identity cannot be a base: affine points carry no infinity flag)");
  return MH_OK;
}

// ---- ark-serialize 0.3 images of G1 points, decoded on the device -------------------------------------------------
// A structured reference string arrives as `Vec<G1Affine>` in ark-serialize's format (kzg10::UniversalParams::powers_of_g,
// what `UniversalSRS::deserialize` reads before `Marlin::index`, /root/reference src/lib.rs:101-113) [ark-serialize /
// ark-ec 0.3, third-party, UPSTREAM-RECALLED; same rules as wire_host.h and oracle/marlin.py g1_compressed]:
//   compressed (CanonicalSerialize::serialize):  x as FQ_B little-endian bytes of the canonical value, flags in the two top
//     bits of the last byte -- bit 7: y > -y (as canonical integers), bit 6: point at infinity;
//   uncompressed (serialize_uncompressed):       x (no flags) followed by y with the infinity flag in bit 6.
// One thread per point: canonical -> Montgomery, y = (x^3 + b)^((p + 1) / 4) (p = 3 mod 4 on both curves), the root with
// the flagged sign; a coordinate >= p, a non-residue, an inconsistent flag or the identity (no infinity on this ABI's
// base sets) counts as bad.  ~570 field multiplications per point: 3 M points decode in well under 0.1 s, where a host
// decodes them at tens of microseconds each.
__device__ __forceinline__ bool fq_canonical_lt_mod(const Fq& a) {
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < Fq::N; i++) { const u64 d = (u64)a.v[i] - CurveFqParams::MOD[i] - borrow; borrow = (u32)(d >> 63); }
  return borrow != 0;
}
__device__ __forceinline__ Fq fq_read_le(const uint8_t* p, uint8_t* flags) {
  Fq a;
#pragma unroll
  for (int i = 0; i < Fq::N; i++) a.v[i] = (u32)p[4 * i] | ((u32)p[4 * i + 1] << 8) | ((u32)p[4 * i + 2] << 16) | ((u32)p[4 * i + 3] << 24);
  *flags = (uint8_t)((a.v[Fq::N - 1] >> 24) & 0xC0u);
  a.v[Fq::N - 1] &= 0x3fffffffu;
  return a;
}
// y > -y on canonical integers, i.e. y > (p - 1) / 2, i.e. 2 y > p
__device__ __forceinline__ bool fq_is_positive(const Fq& y_mont) {
  const Fq y = ff_from_mont(y_mont);
  u32 carry = 0, borrow = 0;
#pragma unroll
  for (int i = 0; i < Fq::N; i++) {
    const u32 twice = (y.v[i] << 1) | carry;
    carry = y.v[i] >> 31;
    const u64 d = (u64)CurveFqParams::MOD[i] - twice - borrow;             // p - 2y
    borrow = (u32)(d >> 63);
  }
  return carry != 0 || borrow != 0;                                        // 2y > p  (2y = p is impossible: p is odd)
}
__global__ __launch_bounds__(128) void g1_deserialize_kernel(const uint8_t* __restrict__ bytes, G1Affine* __restrict__ out, u64 n,
                                                             int compressed, u32* __restrict__ bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int FQB = 4 * Fq::N;
  const uint8_t* p = bytes + i * (u64)(compressed ? FQB : 2 * FQB);
  uint8_t fx, fy = 0;
  Fq x = fq_read_le(p, &fx), y = Fq::zero();
  bool ok = fq_canonical_lt_mod(x);
  if (compressed) {
    ok = ok && !(fx & 0x40u);                                              // the identity cannot be a base
    x = ff_to_mont(x);
    const Fq rhs = ff_add(ff_mul(ff_sqr(x), x), G1_CURVE_B_MONT());
    // e = (p + 1) / 4: p = 3 mod 4, so p + 1 does not carry out of the low two bits' word... computed limb-wise
    u32 e[Fq::N];
    u32 carry = 1;
#pragma unroll
    for (int k = 0; k < Fq::N; k++) { const u64 t = (u64)CurveFqParams::MOD[k] + carry; e[k] = (u32)t; carry = (u32)(t >> 32); }
#pragma unroll
    for (int k = 0; k < Fq::N; k++) e[k] = (e[k] >> 2) | (k + 1 < Fq::N ? e[k + 1] << 30 : carry << 30);
    Fq acc = Fq::one();
    for (int k = Fq::N - 1; k >= 0; k--)
      for (int b = 31; b >= 0; b--) {
        acc = ff_sqr(acc);
        if ((e[k] >> b) & 1u) acc = ff_mul(acc, rhs);
      }
    y = acc;
    ok = ok && ff_sqr(y) == rhs;                                           // otherwise x^3 + b is a non-residue
    if (fq_is_positive(y) != ((fx & 0x80u) != 0)) y = ff_neg(y);
  } else {
    ok = ok && fx == 0;
    y = fq_read_le(p + FQB, &fy);
    ok = ok && fq_canonical_lt_mod(y) && fy == 0;                          // fy = 0x40 would be the identity
    x = ff_to_mont(x); y = ff_to_mont(y);
    ok = ok && ff_sqr(y) == ff_add(ff_mul(ff_sqr(x), x), G1_CURVE_B_MONT());
  }
  if (!ok) { atomicAdd(bad, 1u); x = Fq::zero(); y = Fq::zero(); }
  ff_store(&out[i].x, x);
  ff_store(&out[i].y, y);
}

// --------------------------------------------------------------------------------
// MSM driver
// --------------------------------------------------------------------------------
static int msm_set_attrs(Context& c) {
  if (c.msm_attr_done) return MH_OK;
  int lds = 32768 * 4;
  MH_HIP(hipFuncSetAttribute((const void*)msm::hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  MH_HIP(hipFuncSetAttribute((const void*)msm::scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#define MH_PSPLIT_WIDTHS(X) X(14) X(15) X(16) X(17) X(18) X(19) X(20)      /* the widths mh_bases_precompute picks for 2^14 ... 2^22 points */
  MH_HIP(hipFuncSetAttribute((const void*)msmfb::psplit_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
#define MH_PSPLIT_ATTR(C) MH_HIP(hipFuncSetAttribute((const void*)msmfb::psplit_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  MH_PSPLIT_WIDTHS(MH_PSPLIT_ATTR)
#undef MH_PSPLIT_ATTR
  MH_HIP(hipFuncSetAttribute((const void*)msmfb::scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  MH_HIP(hipFuncSetAttribute((const void*)msmfb::plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(msmfb::PLANE_THREADS * sizeof(msmfb::G1Xyzz30))));
  c.msm_attr_done = true;
  return MH_OK;
}

// window sums (device XYZZ) -> one Jacobian point on the host
static HG1 combine_windows(const std::vector<uint64_t>& win, const msm::Plan& pl) {
  HG1 acc = HG1::identity();
  for (int w = (int)pl.W - 1; w >= 0; w--) {
    // acc holds sum_{w' > w} 2^(start[w'] - start[w+1]) * S_w'; shift by the width of window w
    for (uint32_t k = 0; k < pl.win.bits[w]; k++) acc = acc.dbl();
    const uint64_t* p = win.data() + (size_t)w * XYZZ_L;
    HFq X, Y, ZZ, ZZZ;
    memcpy(X.v, p, FQ_B); memcpy(Y.v, p + FQ_L, FQ_B); memcpy(ZZ.v, p + 2 * FQ_L, FQ_B); memcpy(ZZZ.v, p + 3 * FQ_L, FQ_B);
    acc = acc.add(HG1::from_xyzz(X, Y, ZZ, ZZZ));
  }
  return acc;
}

// Pair-tree accumulation (msm_tree.cuh): leaves one XYZZ point per (job, window, bucket) in c.msm_buckets.
static int msm_tree_accumulate(Context& c, const msm::Jobs& jobs, u64 WN, const msm::Plan& p) {
  namespace T = msmtree;
  hipStream_t s = c.stream;
  const u32 WT = p.W * jobs.njobs;
  const u64 NB = (u64)WT * p.nb;
  MH_TRY(c.tr_off[0].ensure((NB + 1) * 4)); MH_TRY(c.tr_off[1].ensure((NB + 1) * 4)); MH_TRY(c.tr_off[2].ensure((NB + 1) * 4));
  MH_TRY(c.tr_cnt[0].ensure(NB * 4)); MH_TRY(c.tr_cnt[1].ensure(NB * 4));
  MH_TRY(c.tr_sums.ensure((NB / 1024 + 8) * 4 + 64));
  const u64 E1max = WN / 2 + NB + 1, E2max = WN / 4 + NB + 1;
  MH_TRY(c.tr_p[0].ensure(E1max * PT_B)); MH_TRY(c.tr_p[1].ensure(E2max * PT_B));
  MH_TRY(c.tr_ob.ensure(E1max * 4)); MH_TRY(c.tr_pre.ensure(E1max * FQ_B));
  u32* d_scal = (u32*)((char*)c.tr_sums.ptr + (NB / 1024 + 8) * 4);       // [0] scan total, [1] max bucket size
  MH_HIP(hipMemsetAsync(d_scal, 0, 16, s));
  u32* ioff = (u32*)c.tr_off[2].ptr;
  hipLaunchKernelGGL(T::init_kernel, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, s, ioff, (const u32*)c.msm_base.ptr,
                     (const u32*)c.msm_tot.ptr, jobs, p.nb, p.W, NB, d_scal + 1);
  u32 maxcnt = 0;
  MH_HIP(hipMemcpyAsync(&maxcnt, d_scal + 1, 4, hipMemcpyDeviceToHost, s));
  MH_HIP(hipStreamSynchronize(s));
  const u32* icnt = (const u32*)c.msm_tot.ptr;
  const G1Affine* pin = nullptr;
  int rounds = 0;
  while ((1u << rounds) < maxcnt) rounds++;
  for (int r = 0; r < rounds; r++) {
    u32* ocnt = (u32*)c.tr_cnt[r & 1].ptr;
    u32* ooff = (u32*)c.tr_off[r & 1].ptr;
    u32* sums = (u32*)c.tr_sums.ptr;
    const u64 nblk = (NB + 1023) / 1024;
    hipLaunchKernelGGL(T::next_counts_kernel, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, s, ocnt, icnt, NB);
    hipLaunchKernelGGL(T::scan_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const u32*)ocnt, sums, NB);
    hipLaunchKernelGGL(T::scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, nblk, d_scal);
    hipLaunchKernelGGL(T::scan_apply_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const u32*)ocnt, (const u32*)sums, ooff, NB,
                       (const u32*)d_scal);
    u32 E = 0;
    MH_HIP(hipMemcpyAsync(&E, d_scal, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    if (E == 0) break;
    T::Round rd;
    rd.jobs = jobs; rd.wnb = p.W * p.nb; rd.sorted = (const u32*)c.msm_sorted.ptr; rd.pin = pin;
    rd.ioff = ioff; rd.icnt = icnt; rd.ooff = ooff; rd.NB = NB; rd.E = E; rd.first = (r == 0);
    u64 ch = E / (256 * 1024); if (ch < 1) ch = 1; if (ch > 32) ch = 32;
    rd.T = (E + ch - 1) / ch;
    MH_TRY(c.tr_prod.ensure(rd.T * FQ_B)); MH_TRY(c.tr_scr.ensure(rd.T * FQ_B));
    G1Affine* pout = (G1Affine*)c.tr_p[r & 1].ptr;
    if ((u64)E * PT_B > c.tr_p[r & 1].cap) return fail(MH_ENOMEM, "msm tree: output buffer too small");
    unsigned g = (unsigned)((rd.T + T::TPB - 1) / T::TPB);
    hipLaunchKernelGGL(T::fwd_kernel, dim3(g), dim3(T::TPB), 0, s, rd, (u32*)c.tr_ob.ptr, (Fq*)c.tr_pre.ptr, (Fq*)c.tr_prod.ptr);
    hipLaunchKernelGGL(T::inv_kernel, dim3((unsigned)((rd.T + T::INV_CH * 64 - 1) / (T::INV_CH * 64))), dim3(64), 0, s,
                       (Fq*)c.tr_prod.ptr, (Fq*)c.tr_scr.ptr, rd.T);
    hipLaunchKernelGGL(T::bwd_kernel, dim3(g), dim3(T::TPB), 0, s, rd, (const u32*)c.tr_ob.ptr, (const Fq*)c.tr_pre.ptr,
                       (const Fq*)c.tr_prod.ptr, pout);
    MH_HIP(hipGetLastError());
    ioff = ooff; icnt = ocnt; pin = pout;
  }
  if (pin)
    hipLaunchKernelGGL(T::to_buckets_kernel, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, s, (G1Xyzz*)c.msm_buckets.ptr, pin,
                       (const u32*)ioff, icnt, NB);
  else
    hipLaunchKernelGGL(T::to_buckets0_kernel, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, s, (G1Xyzz*)c.msm_buckets.ptr,
                       jobs, p.W * p.nb, (const u32*)c.msm_sorted.ptr, (const u32*)ioff, icnt, NB);
  MH_HIP(hipGetLastError());
  return MH_OK;
}


// --------------------------------------------------------------------------------
// Fixed-base path (msm_fb.cuh): all jobs of the group read one precomputed window table
// --------------------------------------------------------------------------------
// the tabled base set that contains [b, b + n) (callers pass plain device pointers into uploaded base sets)
static const BaseSet* find_table(Context& c, const void* b, size_t n, size_t& off) {
  for (auto& kv : c.bases) {
    const BaseSet& bs = kv.second;
    if (!bs.d_table) continue;
    const char* lo = (const char*)bs.d_points;
    const char* p = (const char*)b;
    if (p >= lo && p + n * PT_B <= lo + bs.n * PT_B && (size_t)(p - lo) % PT_B == 0) { off = (size_t)(p - lo) / PT_B; return &bs; }
  }
  return nullptr;
}

// One sub-batch of a group of jobs on the fixed-base path: everything the stages need, so that its sort, its
// accumulation and its bucket reduction can be issued separately (and on different streams, see msm_fb_pipeline).
// shard = {rank, world}: this rank accumulates and reduces only the buckets of the partitions v = rank (mod world) and the
// results are its PARTIAL sums (partial = true); when the table has fewer partitions than ranks the group is not sharded.
struct FbRun {
  Context& c; const BaseSet& bs; Context::FbWs& ws;
  int nj = 0, is_mont = 0;
  std::vector<size_t> offs, ns, strides; std::vector<const void*> sc;
  msm::Windows win; u32 W = 0, nbt = 0, pshift = 0, nb = 0, nparts = 0, WT = 0; size_t WB = 0;
  msmfb::Own own{0, 1}; bool partial = false; u32 nbown = 0, S = 0;
  msmfb::FbJobs jobs; std::vector<int> alias;
  u64 ent = 0, lsto = 0, max_tiles_total = 0; u32 max_blk = 0, SW = 0;
  std::vector<u32> bpt;                            // split blocks per hist / scatter tile, per partition
  msmfb::RsPlan rs; std::vector<u64> coef;        // bucket reduction: matrix shape, chunking, planes; the host's coefficient of each plane
  std::vector<msmfb::FbWin> desc; std::vector<msmfb::FbBlk> blk;
  bool vmode = false, vcut = false; u32 vT0 = 0, vTs0 = 0, vcap = 0;       // thin launch: buckets accumulated in parts over virtual slots (msm_fb.cuh: VTab)
  bool skewed = false;
  u32 skew_limit = 0xffffffffu, skew_floor = 4096;   // a batch is skewed when its largest bucket exceeds max(skew_floor, 32 x the average)
  FbRun(Context& c_, const BaseSet& bs_, Context::FbWs& ws_) : c(c_), bs(bs_), ws(ws_) {}

  int prepare(int is_mont_, const int* shard) {
    namespace F = msmfb;
    nj = (int)ns.size(); is_mont = is_mont_;
    W = msm::make_windows(bs.tab_c, win);
    nbt = 1u << (bs.tab_c - 1);                                          // buckets per job
    pshift = F::part_bits(bs.tab_c);
    nb = 1u << pshift;                                                   // per virtual window
    nparts = nbt / nb;
    WT = nparts * nj;
    WB = (size_t)nbt * nj;
    own = F::Own{0, 1}; partial = false;
    if (shard && shard[1] > 1 && nparts >= (u32)shard[1]) { own.first = (u32)shard[0]; own.stride = (u32)shard[1]; partial = true; }
    const u32 nown = (nparts - own.first + own.stride - 1) / own.stride; // owned partitions of every job
    nbown = nown * nb;                                                   // owned buckets of every job
    S = F::split_scalars(W);                                             // scalars per count / split block
    memset(&jobs, 0, sizeof(jobs));
    jobs.njobs = nj;
    // A job whose scalars ARE another job's (same device vector, same length) with bases further into the same set --
    // MarlinKZG10 commits a degree-bounded polynomial against powers and against shifted_powers(d) -- has the same digits,
    // hence the same sorted bucket lists: it skips the sort stages and reads the other job's lists with its table indices
    // shifted by the distance of the base ranges (FbWin::delta).  Accumulation and reduction stay per job.
    alias.assign(nj, -1);
    for (int k = 1; k < nj; k++)
      for (int j = 0; j < k; j++)
        if (alias[j] < 0 && sc[j] == sc[k] && ns[j] == ns[k] && offs[k] >= offs[j] && strides[j] == strides[k]) { alias[k] = j; break; }
    ent = 0; lsto = 0; max_blk = 0;
    SW = S * W;
    for (int k = 0; k < nj; k++) {
      jobs.scalars[k] = (const Fr*)sc[k]; jobs.n[k] = ns[k]; jobs.tab_off[k] = (u32)offs[k]; jobs.tab_stride[k] = (u32)strides[k];
      jobs.ent_off[k] = ent; jobs.lst_off[k] = lsto;
      if (alias[k] >= 0) { jobs.nblk[k] = 0; continue; }                              // no blocks: psplit skips the job
      jobs.nblk[k] = (u32)((ns[k] + S - 1) / S);
      ent += (u64)jobs.nblk[k] * SW;                                                  // a split block owns S W entry slots
      lsto += (u64)jobs.nblk[k] * (nparts + 1);
      max_blk = std::max(max_blk, jobs.nblk[k]);
    }
    if (ent >= (1ull << 32)) return fail(MH_EINVAL, "msm batch too large for 32-bit entry offsets");
    // Tiles of the counting sort: a tile = the runs of one virtual window in `bpt` consecutive split blocks, bpt from the load a
    // uniformly distributed scalar puts on the window's partition (so that nothing has to be counted first): window w of wb bits
    // spreads its non-zero digits evenly over the buckets 1 ... 2^(wb - 1), i.e. with density 2 / 2^wb per bucket (the top window a
    // little denser: the scalar is below r, not below 2^256 -- covered by the margin).  A tile that exceeds MAX_TILE all the same --
    // digits that repeat -- is worked off in sub-tiles by the scatter.
    bpt.assign(nparts, 1);
    for (u32 v = 0; v < nparts; v++) {
      double per_block = 0;                      // expected entries of partition v in one split block
      for (u32 w = 0; w < W; w++) {
        const double top = (double)(1ull << (win.bits[w] - 1)), lo = (double)v * nb, hi = std::min((double)(v + 1) * nb, top);
        if (hi > lo) per_block += (double)S * (hi - lo) * 2.0 / (double)(1ull << win.bits[w]);
      }
      const double fit = 0.90 * (double)F::MAX_TILE / std::max(per_block, 1e-9);     // (0.78 / 0.90 / 0.95: 10.25 / 10.04 / 10.05 ms of sort + reduce, profiles/r06h_*)
      bpt[v] = (u32)std::max(1.0, std::min(fit, (double)F::MAX_BPT));
    }
    max_tiles_total = 0;
    for (int k = 0; k < nj; k++)
      for (u32 v = 0; v < nparts && alias[k] < 0; v++)
        if (v % own.stride == own.first) max_tiles_total += (jobs.nblk[k] + bpt[v] - 1) / bpt[v];
    max_tiles_total += 1;
    MH_TRY(ws.dig.ensure(ent * 2 + 16)); MH_TRY(ws.val.ensure(ent * 4 + 16)); MH_TRY(ws.sorted.ensure(ent * 4 + 16));
    MH_TRY(ws.pc.ensure(lsto * 2 + 16)); MH_TRY(ws.ptot.ensure((size_t)WT * 8)); MH_TRY(ws.desc.ensure((size_t)WT * sizeof(msmfb::FbWin)));
    MH_TRY(ws.blk.ensure(8 * max_tiles_total * sizeof(F::FbBlk)));
    MH_TRY(ws.bh.ensure(max_tiles_total * nb * 4));
    MH_TRY(ws.tot.ensure(WB * 4)); MH_TRY(ws.base.ensure(WB * 4)); MH_TRY(ws.pend.ensure(WB * 4));
    MH_TRY(ws.buckets.ensure(WB * sizeof(F::G1Xyzz30)));
    // Bucket reduction (msm_fb.cuh: rsum / plane kernels): the owned buckets as R_own rows of C, every row and every column summed
    // by a group of lanes -- as many lanes that the launch has about two waves per SIMD (the kernel holds 220 registers), at
    // most a wave's 64 --, then the bit planes of the row and column indices.  With a rank's partitions interleaved (v = first, first + stride, ...) the
    // global row of local row m is r(m) = first rpp + stride rpp (m / rpp) + m % rpp: the planes run over the bits of m and
    // the constants go into the host's coefficients.
    {
      memset(&rs, 0, sizeof(rs));
      u32 lgown = 0;
      while ((1ull << lgown) < nbown) lgown++;
      rs.lgC = std::min<u32>(pshift, (lgown + 1) / 2);
      if (rs.lgC < 1) rs.lgC = 1;
      if (rs.lgC > pshift) return fail(MH_EINVAL, "fixed-base MSM: partition narrower than a row of the bucket matrix");   // (window widths >= 4 never get here)
      rs.C = 1u << rs.lgC;
      rs.lgrpp = pshift - rs.lgC;
      rs.R_own = nbown >> rs.lgC;
      rs.lgM = 0;
      while ((1u << rs.lgM) < rs.R_own) rs.lgM++;
      // Buckets per lane: two waves per SIMD on the full launches of a 2^20 proof (sort + reduce 9.34 vs 9.64 ms per proof at one
      // wave), one wave where two would leave a lane fewer than 16 buckets (2^16 + SonicKZG10: 2.63 vs 2.88 ms) -- either way the
      // kernel is bound by VALU issue: one wave with its three interleaved multiplication chains already fills the SIMD
      // (profiles/r05c_sweep_rsum_threads.txt).
      const u64 per_simd = 2ull * nj * nbown;
      const u64 simds = (u64)c.num_simds;
      u64 Lt = (per_simd + 128 * simds - 1) / (128 * simds);
      if (Lt < 16) Lt = std::min<u64>(16, (per_simd + 64 * simds - 1) / (64 * simds));
      if (Lt < 1) Lt = 1;
      auto lanes = [&](u64 len) { u32 lg = 0; while (lg < 6 && (2ull << lg) * Lt <= len) lg++; return lg; };
      rs.lgJ = lanes(rs.C); rs.J = 1u << rs.lgJ; rs.Lr = rs.C >> rs.lgJ;
      rs.lgI = lanes(1ull << rs.lgM); rs.I = 1u << rs.lgI; rs.Lc = (rs.R_own + rs.I - 1) / rs.I;
      rs.NTr = (u32)(((u64)rs.R_own * rs.J + 63) & ~63ull);
      rs.NT = rs.NTr + (u32)(((u64)rs.C * rs.I + 63) & ~63ull);
      rs.NS = rs.R_own + rs.C;
      rs.nplanes = rs.lgC + rs.lgM + 1;
      coef.assign(rs.nplanes, 0);
      for (u32 p = 0; p < rs.lgC; p++) coef[p] = 1ull << p;
      for (u32 p = 0; p < rs.lgM; p++) coef[rs.lgC + p] = ((u64)rs.C << p) * (p < rs.lgrpp ? 1ull : (u64)own.stride);
      coef[rs.nplanes - 1] = (((u64)rs.C * own.first) << rs.lgrpp) + 1;
    }
    MH_TRY(ws.seg.ensure((size_t)nj * rs.NS * sizeof(F::G1Xyzz30)));
    MH_TRY(ws.win.ensure((size_t)nj * rs.nplanes * sizeof(G1Xyzz)));
    MH_TRY(ws.sums.ensure(64 + F::SIZE_BINS * 4));
    MH_TRY(ws.perm.ensure(WB * 4));
    // The accumulate kernel runs over virtual slots (msm_fb.cuh: VTab) -- persistent waves instead of the hardware's dispatch of
    // blocks -- when the launch fills the chip (more than 6 waves of buckets per SIMD) or belongs to a bucket-range shard; a shard
    // also has its buckets above Ts entries accumulated in parts of T: T = half, Ts = two thirds of the entries a lane gets when the
    // launch's entries are spread evenly over the 2 x 64 lanes of every SIMD, for uniformly distributed digits (a rank of 8 at
    // 2^20: 7.31-7.37 ms of accumulation per proof at 50 / 66 % and at 33 / 33 %, 7.5-8.1 at 25 / 50 % and at 50 / 100 %, 8.0 with the
    // hardware's dispatch: profiles/r06u_*), Ts at least 1.25 x the average bucket; the device doubles both until the parts fit
    // the buffer.  Not cut on one GPU: merging the parts costs 7-8 % of the launch at 2^16, as much as the cut buys there, and at
    // 2^20 nothing is above Ts; and the thin launches of a small proof on one GPU (2^16: one round of waves) keep the hardware's
    // dispatch, which is 4 % faster there (profiles/r06z_*).  MH_ACC_PARTS=0: the one-thread-per-bucket kernel everywhere (cross-check).
    {
      static const bool parts_env = [] { const char* e = getenv("MH_ACC_PARTS"); return !e || atoi(e) != 0; }();
      const u64 active = (u64)nj * nbown;
      const u64 per_simd = (active / 64 + (u64)c.num_simds - 1) / (u64)c.num_simds;
      vmode = parts_env && (partial || per_simd > 6);
      vcut = vmode && partial;
      if (vmode) {
        double ent_own = 0;
        for (int k = 0; k < nj; k++) ent_own += (double)ns[k] * W * (double)nbown / (double)nbt;
        const double per_lane = ent_own / (128.0 * (double)c.num_simds);
        const double avg = ent_own / (double)std::max<u64>(active, 1);      // expected entries per owned bucket
        vT0 = (u32)std::max(8.0, per_lane * 0.50);
        vTs0 = vcut ? (u32)std::max(std::max((double)vT0, per_lane * 0.66), 1.25 * avg) : 0x7fffffffu;
        vcap = (u32)(4 * 128 * c.num_simds);           // parts of cut buckets <= entries (1 / T + 1 / Ts) = 3.5 per lane for the expected entries
        MH_TRY(ws.vtab.ensure(sizeof(F::VTab)));
        if (vcut) MH_TRY(ws.aux.ensure((size_t)vcap * sizeof(F::G1Xyzz30)));
      }
    }
    desc.assign(WT, msmfb::FbWin{});
    return MH_OK;
  }

  // psplit / hist / colscan / binscan / woff / scatter + the size order of the buckets: no host round trip (round 6: the one-pass
  // partition writes every split block's entries into the block's own region, so no stage needs a count the host would have to
  // fetch; what the host uploads -- tiles, block list, aliases -- depends on the batch's shape only).  The skew decision travels
  // with the results (finish): skewed = true there means the caller takes the variable-base path instead.
  int sort(hipStream_t s) {
    namespace F = msmfb;
    ProfScope ps(c, PF_MSM_STAGES, s);
    unsigned short* key = (unsigned short*)ws.dig.ptr; u32* val = (u32*)ws.val.ptr;
    unsigned short* lst = (unsigned short*)ws.pc.ptr;
    u32* d_wtot = (u32*)ws.ptot.ptr; u32* d_src = d_wtot + WT;
    // virtual-window descriptors (the host's part: tiles, histogram offsets, alias shifts) and the XCD-interleaved block list; grid
    // = 8 x the busiest XCD's tile count.  The XCD of a window is its rank among the windows that HAVE tiles, mod 8: with the MSM
    // sharded over G ranks a rank owns the partitions v = g (mod G), and numbering by gw would put all of them on one XCD for G = 8
    u64 bho = 0, xcd_tiles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<u32> live, src(WT, 0xffffffffu);
    for (int k = 0; k < nj; k++)
      for (u32 v = 0; v < nparts; v++) {
        const u32 gw = k * nparts + v;
        msmfb::FbWin& d = desc[gw];
        d = msmfb::FbWin{};
        d.bpt = bpt[v]; d.bh_off = bho;
        if (alias[k] >= 0) { src[gw] = alias[k] * nparts + v; d.delta = (u32)(offs[k] - offs[alias[k]]); continue; }   // shares the lists: no tiles of its own
        if (v % own.stride != own.first) continue;                                      // another rank's partition: stays empty
        d.ntiles = (jobs.nblk[k] + bpt[v] - 1) / bpt[v];
        bho += (u64)d.ntiles * nb;
        if (d.ntiles) { xcd_tiles[live.size() & 7] += d.ntiles; live.push_back(gw); }
      }
    u64 grid_tiles = 0;
    for (int x = 0; x < 8; x++) grid_tiles = std::max(grid_tiles, xcd_tiles[x]);
    blk.assign(grid_tiles * 8, F::FbBlk{0xffffffffu, 0});
    for (u32 x = 0; x < 8; x++) {
      u64 k = 0;
      for (size_t li = x; li < live.size(); li += 8)
        for (u32 t = 0; t < desc[live[li]].ntiles; t++) blk[(k++ << 3) | x] = F::FbBlk{live[li], t};
    }
    // the pinned copies are reused by the next group only after this group's finish() has synchronised the stream
    const size_t desc_b = (size_t)WT * sizeof(msmfb::FbWin), src_b = (size_t)WT * 4;
    MH_TRY(ws.h_desc.ensure(desc_b + src_b)); MH_TRY(ws.h_blk.ensure(blk.size() * sizeof(F::FbBlk) + 8));
    memcpy(ws.h_desc.ptr, desc.data(), desc_b); memcpy((char*)ws.h_desc.ptr + desc_b, src.data(), src_b);
    MH_HIP(hipMemcpyAsync(ws.desc.ptr, ws.h_desc.ptr, desc_b, hipMemcpyHostToDevice, s));
    MH_HIP(hipMemcpyAsync(d_src, (char*)ws.h_desc.ptr + desc_b, src_b, hipMemcpyHostToDevice, s));
    if (grid_tiles) {
      memcpy(ws.h_blk.ptr, blk.data(), blk.size() * sizeof(F::FbBlk));
      MH_HIP(hipMemcpyAsync(ws.blk.ptr, ws.h_blk.ptr, blk.size() * sizeof(F::FbBlk), hipMemcpyHostToDevice, s));
    }
    u32* d_max = (u32*)ws.sums.ptr;                    // [0] largest bucket, [1] buckets with deferred entries
    MH_HIP(hipMemsetAsync(d_max, 0, 8, s));
    // the usual widths: recoding unrolled over compile-time windows (msm::for_each_digit_c; the generic kernel -- its scalar's words
    // indexed at run time, i.e. served from scratch -- costs 0.6 ms more per proof on a rank of 8 at 2^20, profiles/r06m_*)
    if (max_blk) {
      switch (bs.tab_c) {
#define MH_PSPLIT_CASE(C) case C: hipLaunchKernelGGL(F::psplit_kernel<C>, dim3(max_blk, nj), dim3(F::SORT_THREADS), F::split_lds_bytes(W), s, jobs, key, val, lst, W, win, \
                                                     is_mont, nparts, pshift, (u32)bs.n, S, own); break;
        MH_PSPLIT_WIDTHS(MH_PSPLIT_CASE)
#undef MH_PSPLIT_CASE
        default: hipLaunchKernelGGL(F::psplit_kernel<0>, dim3(max_blk, nj), dim3(F::SORT_THREADS), F::split_lds_bytes(W), s, jobs, key, val, lst, W, win, is_mont,
                                    nparts, pshift, (u32)bs.n, S, own);
      }
    }
    msmfb::FbWin* fbw = (msmfb::FbWin*)ws.desc.ptr;
    const F::FbBlk* dblk = (const F::FbBlk*)ws.blk.ptr;
    if (grid_tiles)
      hipLaunchKernelGGL(F::hist_kernel, dim3((unsigned)(grid_tiles * 8)), dim3(F::SORT_THREADS), F::hist_lds_bytes(nb), s, (const msmfb::FbWin*)fbw, dblk, jobs,
                         (const unsigned short*)key, (const unsigned short*)lst, (u32*)ws.bh.ptr, nb, nparts, SW);
    hipLaunchKernelGGL(F::colscan_kernel, dim3((nb + 255) / 256, WT), dim3(256), 0, s, (const msmfb::FbWin*)fbw, (u32*)ws.bh.ptr, (u32*)ws.tot.ptr, nb);
    hipLaunchKernelGGL(msm::binscan_kernel, dim3(WT), dim3(1024), 0, s, (const u32*)ws.tot.ptr, (u32*)ws.base.ptr, nb, d_max, d_wtot);
    hipLaunchKernelGGL(F::woff_kernel, dim3(1), dim3(1024), 0, s, fbw, (const u32*)d_wtot, (const u32*)d_src, WT);
    for (int k = 0; k < nj; k++)
      if (alias[k] >= 0) {          // an aliasing job has the bucket sizes and list positions of the job whose lists it reads
        MH_HIP(hipMemcpyAsync((u32*)ws.tot.ptr + (size_t)k * nbt, (const u32*)ws.tot.ptr + (size_t)alias[k] * nbt, (size_t)nbt * 4, hipMemcpyDeviceToDevice, s));
        MH_HIP(hipMemcpyAsync((u32*)ws.base.ptr + (size_t)k * nbt, (const u32*)ws.base.ptr + (size_t)alias[k] * nbt, (size_t)nbt * 4, hipMemcpyDeviceToDevice, s));
      }
    if (grid_tiles)
      hipLaunchKernelGGL(F::scatter_kernel, dim3((unsigned)(grid_tiles * 8)), dim3(F::SORT_THREADS), F::scatter_lds_bytes(nb), s, (const msmfb::FbWin*)fbw, dblk, jobs,
                         (const unsigned short*)key, (const u32*)val, (const unsigned short*)lst, (const u32*)ws.bh.ptr, (const u32*)ws.base.ptr,
                         (const u32*)ws.tot.ptr, (u32*)ws.sorted.ptr, nb, nparts, SW);
    // buckets ordered by size, largest first (before the skew decision comes back: a few tens of microseconds, wasted
    // only on the skewed batches that leave this path anyway)
    u32* d_szh = (u32*)ws.sums.ptr + 16;
    MH_HIP(hipMemsetAsync(d_szh, 0, F::SIZE_BINS * 4, s));
    const u64 OW = (u64)nj * nbown;                    // the owned buckets of all jobs: what is ordered, and what the accumulate kernel is launched over
    hipLaunchKernelGGL(F::size_hist_kernel, dim3((unsigned)((OW + 1023) / 1024)), dim3(1024), 0, s, (const u32*)ws.tot.ptr, OW, nbown, nbt, pshift, own, d_szh);
    if (vmode) hipLaunchKernelGGL(F::size_vscan_kernel, dim3(1), dim3(1024), 0, s, d_szh, (F::VTab*)ws.vtab.ptr, vT0, vTs0, vcap);
    else hipLaunchKernelGGL(F::size_scan_kernel, dim3(1), dim3(1024), 0, s, d_szh);
    hipLaunchKernelGGL(F::size_perm_kernel, dim3((unsigned)((OW + 1023) / 1024)), dim3(1024), 0, s, (const u32*)ws.tot.ptr, OW, nbown, nbt, pshift, own, d_szh,
                       (u32*)ws.perm.ptr);
    // The skew decision (largest bucket > max(4096, 32 x average)) is taken on the device by the accumulate kernel and read
    // by the host together with the results (finish): no host round trip between the sort and the accumulation.
    u64 pairs = 0;
    for (int k = 0; k < nj; k++) pairs += ns[k];
    const u64 avg = pairs * W / WB + 1;
    skew_limit = (u32)std::min<u64>(std::max<u64>(skew_floor, 32 * avg), 0xffffffffull);
    // (a strided slice has no variable-base fallback -- its bases are not a contiguous range --: msm_batch_strided_device turns a
    // skewed batch into an error instead of letting one thread walk a list of millions)
    return MH_OK;
  }

  int accum(hipStream_t s) {
    namespace F = msmfb;
    const msmfb::FbWin* fbw = (const msmfb::FbWin*)ws.desc.ptr;
    u32* d_max = (u32*)ws.sums.ptr;
    {
      ProfScope pa(c, PF_MSM_ACCUM, s);
      const u64 OW = (u64)nj * nbown;
      // (MH_ACC_PARTS=0, the cross-check: one thread per bucket, the hardware's dispatch.  Resident waves per SIMD there: 3 with the
      // chip full, 2 -- no spills, no lone fourth wave -- when a launch has at most 6 waves per SIMD: 8.2 / 9.1 / 10.5 ms of
      // accumulation per proof on a simulated rank of 8 at 2 / 3 / 4 waves, profiles/r03j_sim_*.)
      const u64 active = (u64)nj * nbown;                                     // buckets that do work on this rank
      const u64 per_simd = (active / 64 + (u64)c.num_simds - 1) / (u64)c.num_simds;
      const int acc_waves = per_simd <= 6 ? 2 : 3;      // (blocks of 64 or 128 threads instead of 256 on the thin launches: no difference, profiles/r06m_*)
      const unsigned tpb = (unsigned)msm::ACC_TPB;
      const u64 nblk = (OW + tpb - 1) / tpb;
      if (vmode) {
        MH_HIP(hipMemsetAsync(ws.pend.ptr, 0, WB * 4, s));
        const unsigned grid = (unsigned)(2 * c.num_simds * 64 / tpb);            // two waves per SIMD, resident for the whole launch
        hipLaunchKernelGGL(F::accum30v_kernel, dim3(grid), dim3(tpb), 0, s, fbw, (const F::G1Aff30*)bs.d_table,
                           (const u32*)ws.sorted.ptr, (const u32*)ws.base.ptr, (const u32*)ws.tot.ptr, (const u32*)ws.perm.ptr,
                           (F::G1Xyzz30*)ws.buckets.ptr, (F::G1Xyzz30*)ws.aux.ptr, (u32*)ws.pend.ptr, d_max + 1, nb, (F::VTab*)ws.vtab.ptr,
                           (const u32*)d_max, skew_limit);
        if (vcut)
        hipLaunchKernelGGL(F::merge_parts_kernel, dim3((unsigned)std::min<u64>((OW + 127) / 128, 2048)), dim3(128), 0, s, (const F::VTab*)ws.vtab.ptr,
                           (const u32*)ws.perm.ptr, (const F::G1Xyzz30*)ws.aux.ptr, (F::G1Xyzz30*)ws.buckets.ptr, (const u32*)d_max, skew_limit);
      } else if (acc_waves == 2)
        hipLaunchKernelGGL(F::accum30_kernel<2>, dim3((unsigned)nblk), dim3(tpb), 0, s, fbw, (const F::G1Aff30*)bs.d_table,
                           (u32*)ws.sorted.ptr, (const u32*)ws.base.ptr, (const u32*)ws.tot.ptr, (const u32*)ws.perm.ptr,
                           (F::G1Xyzz30*)ws.buckets.ptr, (u32*)ws.pend.ptr, d_max + 1, nb, OW, (const u32*)d_max, skew_limit);
      else
        hipLaunchKernelGGL(F::accum30_kernel<3>, dim3((unsigned)nblk), dim3(tpb), 0, s, fbw, (const F::G1Aff30*)bs.d_table,
                           (u32*)ws.sorted.ptr, (const u32*)ws.base.ptr, (const u32*)ws.tot.ptr, (const u32*)ws.perm.ptr,
                           (F::G1Xyzz30*)ws.buckets.ptr, (u32*)ws.pend.ptr, d_max + 1, nb, OW, (const u32*)d_max, skew_limit);
    }
    ProfScope ps(c, PF_MSM_STAGES, s);
    hipLaunchKernelGGL(F::fixup30_kernel, dim3((unsigned)std::min<u64>((WB + 63) / 64, 1024)), dim3(64), 0, s, fbw,
                       (const F::G1Aff30*)bs.d_table, (const u32*)ws.sorted.ptr, (const u32*)ws.base.ptr, (const u32*)ws.tot.ptr,
                       (u32*)ws.pend.ptr, (const u32*)(d_max + 1), (F::G1Xyzz30*)ws.buckets.ptr, nb, (u64)WB);
    MH_HIP(hipGetLastError());
    // the caller's independent work (Context::side_job) goes to stream2 behind the accumulation: beside the reduction that follows
    if (c.side_job && s == c.stream && c.stream2 && c.side_ev[0] && c.side_ev[1]) {
      MH_HIP(hipEventRecord(c.side_ev[0], s));
      MH_HIP(hipStreamWaitEvent(c.stream2, c.side_ev[0], 0));
      std::function<int()> job;
      job.swap(c.side_job);                          // consumed, whatever it returns
      hipStream_t main_stream = c.stream;
      c.stream = c.stream2;
      c.prof_override = PF_SIDE;
      const int rc = job();
      c.prof_override = -1;
      c.stream = main_stream;
      MH_HIP(hipEventRecord(c.side_ev[1], c.stream2));
      c.side_ran = true;
      MH_TRY(rc);
    }
    return MH_OK;
  }

  // one bucket set of nbt buckets per job: bucket b (0-based, across the virtual windows) weighs b + 1.  Row / column sums, then the
  // bit planes of the row and column indices (msm_fb.cuh); finish() combines the planes on the host.
  int reduce(hipStream_t s) {
    namespace F = msmfb;
    ProfScope ps(c, PF_MSM_STAGES, s);
    ProfScope pr(c, PF_MSM_REDUCE, s);
    F::G1Xyzz30* sums = (F::G1Xyzz30*)ws.seg.ptr;
    const u32* d_max = (const u32*)ws.sums.ptr;
    const u64 threads = (u64)nj * rs.NT;
    hipLaunchKernelGGL(F::rsum_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const F::G1Xyzz30*)ws.buckets.ptr, sums, nbt,
                       (u32)nj, rs, own, d_max, skew_limit);
    hipLaunchKernelGGL(F::plane_kernel, dim3(rs.nplanes, (unsigned)nj), dim3(F::PLANE_THREADS), F::PLANE_THREADS * sizeof(F::G1Xyzz30), s,
                       (const F::G1Xyzz30*)sums, (G1Xyzz*)ws.win.ptr, rs, d_max, skew_limit);
    MH_HIP(hipGetLastError());
    return MH_OK;
  }

  // the planes of every job to the host (synchronises `s`), then sum_planes coef[p] x plane[p] per job: one shared chain of
  // doublings with the planes added where their coefficient has the bit -- ~20 doublings and ~20 additions of a CPU core, the
  // jobs side by side on the host pool
  int finish(hipStream_t s, HG1* out) {
    const size_t npl = rs.nplanes, sums_n = (size_t)nj * npl * XYZZ_L;
    MH_TRY(ws.h_out.ensure(sums_n * 8 + 8));
    uint64_t* sums_h = (uint64_t*)ws.h_out.ptr;
    MH_HIP(hipMemcpyAsync(sums_h, ws.win.ptr, sums_n * 8, hipMemcpyDeviceToHost, s));
    MH_HIP(hipMemcpyAsync(sums_h + sums_n, ws.sums.ptr, 4, hipMemcpyDeviceToHost, s));
    mh::host_tick("finish: before sync");
    MH_HIP(hipStreamSynchronize(s));
    mh::host_tick("finish: synced (device done)");
    const u32 mx = *(const u32*)(sums_h + sums_n);
    skewed = mx > skew_limit;                      // the kernels returned at once: `out` is not a result
    if (skewed) return MH_OK;
    int top = 0;
    for (u64 cf : coef) while ((cf >> top) > 1) top++;
    const std::vector<u64>* cfs = &coef;
    auto combine = [sums_h, npl, top, cfs](int k) {
      std::vector<HG1> pl(npl);
      for (size_t i = 0; i < npl; i++) {
        const uint64_t* p = sums_h + ((size_t)k * npl + i) * XYZZ_L;
        HFq X, Y, ZZ, ZZZ;
        memcpy(X.v, p, FQ_B); memcpy(Y.v, p + FQ_L, FQ_B); memcpy(ZZ.v, p + 2 * FQ_L, FQ_B); memcpy(ZZZ.v, p + 3 * FQ_L, FQ_B);
        pl[i] = HG1::from_xyzz(X, Y, ZZ, ZZZ);
      }
      HG1 acc = HG1::identity();
      for (int bit = top; bit >= 0; bit--) {
        acc = acc.dbl();
        for (size_t i = 0; i < npl; i++) if (((*cfs)[i] >> bit) & 1) acc = acc.add(pl[i]);
      }
      return acc;
    };
    std::vector<std::future<HG1>> fut(nj);
    WaitAll wait; 
    const bool pool = !(c.diag & 2u);
    for (int k = 1; k < nj && pool; k++) fut[k] = host_pool().submit([combine, k] { return combine(k); });
    wait.add(fut);
    out[0] = combine(0);
    for (int k = 1; k < nj; k++) out[k] = pool ? fut[k].get() : combine(k);
    mh::host_tick("finish: planes combined");
    return check_result(s, out);
  }

  // ---- MH_CHECK: the invariants of msm_check.cuh, stage by stage (Context::Check) --------------------------------------------
  bool chk_skip = false;                         // the batch left this path on the device (skew): nothing after the sort holds results
  std::string chk_log;
  int chk_fail(const std::string& stage, const std::string& what) {
    c.chk.violations++;
    chk_log += stage + ": VIOLATED: " + what + "\n";
    c.chk.report = chk_log;
    return fail(MH_ECHECK, "MH_CHECK: fixed-base MSM batch #" + std::to_string(c.chk.batches) + ", stage " + stage + ": " + what);
  }
  std::string chk_shape() const {
    char b[256];
    snprintf(b, sizeof(b), "batch #%llu: jobs %d, c %u, windows %u, partitions %u x %u buckets, own {%u, %u}, entries %llu, strided %d",
             (unsigned long long)c.chk.batches, nj, bs.tab_c, W, nparts, nb, own.first, own.stride, (unsigned long long)ent,
             (int)(!strides.empty() && strides[0] != 1));
    return b;
  }
  // after sort(): the lists against the scalars
  int check_sort(hipStream_t s) {
    namespace K = msmchk;
    c.chk.batches++;
    chk_log = chk_shape() + "\n";
    const size_t sums_b = 3 * (size_t)WT * sizeof(K::Sum), tot_b = 2 * (size_t)WT * 4, bad_b = 64, desc_b = (size_t)WT * sizeof(msmfb::FbWin);
    MH_TRY(c.chk.d.ensure(sums_b + tot_b + bad_b)); MH_TRY(c.chk.h.ensure(sums_b + tot_b + bad_b + desc_b + 8));
    K::Sum* d_sum = (K::Sum*)c.chk.d.ptr; u32* d_tot = (u32*)((char*)c.chk.d.ptr + sums_b); u32* d_bad = d_tot + 2 * WT;
    MH_HIP(hipMemsetAsync(c.chk.d.ptr, 0, sums_b + tot_b + bad_b, s));
    u64 nmax = 0;
    for (int k = 0; k < nj; k++) nmax = std::max<u64>(nmax, ns[k]);
    const unsigned gx = (unsigned)std::min<u64>((nmax + 255) / 256, 2048);
    hipLaunchKernelGGL(K::recode_kernel, dim3(gx, nj), dim3(256), 0, s, jobs, d_sum, W, win, is_mont, nparts, pshift, (u32)bs.n, own);
    const msmfb::FbWin* fbw = (const msmfb::FbWin*)ws.desc.ptr;
    if (max_blk)
      hipLaunchKernelGGL(K::runs_kernel, dim3(max_blk, nj), dim3(256), 0, s, jobs, (const u32*)ws.val.ptr, (const unsigned short*)ws.pc.ptr, d_sum + WT, nparts, SW, d_bad + 8);
    hipLaunchKernelGGL(K::entries_kernel, dim3(64, WT), dim3(256), 0, s, fbw, (const u32*)ws.sorted.ptr, d_sum + 2 * WT);
    hipLaunchKernelGGL(K::tot_kernel, dim3(WT), dim3(64), 0, s, (const u32*)ws.tot.ptr, (const u32*)ws.base.ptr, nb, d_tot);
    if (c.chk.level >= 2)
      hipLaunchKernelGGL(K::lists_kernel, dim3((unsigned)((WB + 127) / 128)), dim3(128), 0, s, fbw, jobs, (const u32*)ws.sorted.ptr, (const u32*)ws.base.ptr,
                         (const u32*)ws.tot.ptr, nb, (u64)WB, nparts, pshift, W, win, is_mont, (u32)bs.n, d_bad);
    MH_HIP(hipGetLastError());
    char* h = (char*)c.chk.h.ptr;
    MH_HIP(hipMemcpyAsync(h, c.chk.d.ptr, sums_b + tot_b + bad_b, hipMemcpyDeviceToHost, s));
    MH_HIP(hipMemcpyAsync(h + sums_b + tot_b + bad_b, ws.desc.ptr, desc_b, hipMemcpyDeviceToHost, s));
    MH_HIP(hipMemcpyAsync(h + sums_b + tot_b + bad_b + desc_b, ws.sums.ptr, 4, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    const K::Sum* hs = (const K::Sum*)h; const u32* ht = (const u32*)(h + sums_b); const u32* hb = ht + 2 * WT;
    const msmfb::FbWin* hd = (const msmfb::FbWin*)(h + sums_b + tot_b + bad_b);
    const u32 largest = *(const u32*)(h + sums_b + tot_b + bad_b + desc_b);
    chk_skip = largest > skew_limit;
    if (hb[8]) return chk_fail("split", std::to_string(hb[8]) + " split block(s) wrote a malformed run table (boundaries not non-decreasing from 0, or past the block's region)");
    u64 total = 0, off = 0;
    for (u32 gw = 0; gw < WT; gw++) {
      const int k = (int)(gw / nparts);
      if (alias[k] >= 0) {
        const u32 sg = (u32)alias[k] * nparts + gw % nparts;
        if (hd[gw].off != hd[sg].off || hd[gw].cnt != hd[sg].cnt) return chk_fail("descriptor", "job " + std::to_string(k) + " does not point at the lists of job " + std::to_string(alias[k]) + " it shares");
        continue;
      }
      const K::Sum &want = hs[gw], &sp = hs[WT + gw], &sc = hs[2 * WT + gw];
      auto str = [](const K::Sum& x) { char b[96]; snprintf(b, sizeof(b), "(n %llu, sum %llu, xor %08x)", x.cnt, x.sum, x.x); return std::string(b); };
      const std::string where = "job " + std::to_string(k) + " partition " + std::to_string(gw % nparts);
      if (sp.cnt != want.cnt || sp.sum != want.sum || sp.x != want.x) return chk_fail("split", where + ": entries written " + str(sp) + " != entries the scalars give " + str(want));
      if (hd[gw].cnt != want.cnt) return chk_fail("descriptor", where + ": the device put " + std::to_string(hd[gw].cnt) + " entries into the window's descriptor, the scalars have " + std::to_string(want.cnt) + " owned non-zero digits there");
      if (hd[gw].off != off) return chk_fail("descriptor", where + ": list offset " + std::to_string(hd[gw].off) + " is not the sum of the windows before it (" + std::to_string(off) + ")");
      off += want.cnt;
      if (sc.cnt != want.cnt || sc.sum != want.sum || sc.x != want.x) return chk_fail("scatter", where + ": sorted lists " + str(sc) + " are not a permutation of the split's entries " + str(sp));
      if (ht[2 * gw] != want.cnt) return chk_fail("bucket sizes", where + ": sizes sum to " + std::to_string(ht[2 * gw]) + ", the window has " + std::to_string(want.cnt) + " entries");
      if (ht[2 * gw + 1] != 0) return chk_fail("bucket starts", where + ": " + std::to_string(ht[2 * gw + 1]) + " bucket(s) do not start at the exclusive scan of the sizes");
      total += want.cnt;
    }
    if (c.chk.level >= 2 && !chk_skip && hb[0] != 0) {
      char b[160]; snprintf(b, sizeof(b), "%u list entr(ies) recode to another bucket or sign; first: bucket slot %u, entry %08x", hb[0], hb[1], hb[2]);
      return chk_fail("lists", b);
    }
    chk_log += "sort: ok (" + std::to_string(total) + " entries: scalars = split = sorted per partition; descriptors, sizes and starts consistent" +
               (c.chk.level >= 2 ? "; every entry recodes to its bucket" : "") + (chk_skip ? "; SKEWED: the batch leaves this path" : "") + ")\n";
    // what the accumulation and the reduction are about to write starts as garbage that is no point of the curve
    if (!chk_skip) {
      MH_HIP(hipMemsetAsync(ws.buckets.ptr, 0xA5, WB * sizeof(msmfb::G1Xyzz30), s)); MH_HIP(hipMemsetAsync(ws.pend.ptr, 0xA5, WB * 4, s));
      MH_HIP(hipMemsetAsync(ws.seg.ptr, 0xA5, (size_t)nj * rs.NS * sizeof(msmfb::G1Xyzz30), s));
      MH_HIP(hipMemsetAsync(ws.win.ptr, 0xA5, (size_t)nj * rs.nplanes * sizeof(G1Xyzz), s));
    }
    c.chk.report = chk_log;
    return MH_OK;
  }
  // after accum(): the buckets
  int check_accum(hipStream_t s) {
    namespace K = msmchk;
    if (chk_skip) return MH_OK;
    u32* d_bad = (u32*)c.chk.d.ptr;
    MH_HIP(hipMemsetAsync(d_bad, 0, 64, s));
    hipLaunchKernelGGL(K::buckets_kernel, dim3((unsigned)((WB + 63) / 64)), dim3(64), 0, s, (const msmfb::FbWin*)ws.desc.ptr, (const msmfb::G1Aff30*)bs.d_table,
                       (const u32*)ws.sorted.ptr, (const u32*)ws.base.ptr, (const u32*)ws.tot.ptr, (const msmfb::G1Xyzz30*)ws.buckets.ptr,
                       (const u32*)ws.pend.ptr, nb, (u64)WB, nparts, own, c.chk.level >= 2 ? 1 : 0, d_bad);
    MH_HIP(hipGetLastError());
    u32* hb = (u32*)c.chk.h.ptr;
    MH_HIP(hipMemcpyAsync(hb, d_bad, 64, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    auto where = [&](u32 gid) { return "job " + std::to_string(gid / nbt) + " bucket " + std::to_string(gid % nbt); };
    if (hb[0]) return chk_fail("accumulate", std::to_string(hb[0]) + " owned bucket(s) are neither the identity nor on the curve; first: " + where(hb[1]));
    if (hb[2]) return chk_fail("fix-up", std::to_string(hb[2]) + " owned bucket(s) still pending after the fix-up");
    if (hb[3]) return chk_fail("accumulate", std::to_string(hb[3]) + " bucket(s) are not the sum of their list (32-bit complete law); first: " + where(hb[4]));
    chk_log += std::string("accumulate: ok (every owned bucket on the curve, none pending") + (c.chk.level >= 2 ? "; every bucket equals the recomputed sum of its list" : "") + ")\n";
    c.chk.report = chk_log;
    return MH_OK;
  }
  // after reduce(): row / column sums and planes
  int check_reduce(hipStream_t s) {
    namespace K = msmchk;
    if (chk_skip) return MH_OK;
    const size_t rc_b = (size_t)2 * nj * sizeof(G1Xyzz);
    MH_TRY(c.chk.d.ensure(128 + rc_b)); MH_TRY(c.chk.h.ensure(128 + rc_b));
    u32* d_bad = (u32*)c.chk.d.ptr; G1Xyzz* d_rc = (G1Xyzz*)((char*)c.chk.d.ptr + 128);
    MH_HIP(hipMemsetAsync(d_bad, 0, 128, s));
    const u64 nsum = (u64)nj * rs.NS, npl = (u64)nj * rs.nplanes;
    hipLaunchKernelGGL(K::sums_kernel, dim3((unsigned)((nsum + 63) / 64)), dim3(64), 0, s, (const msmfb::G1Xyzz30*)ws.seg.ptr, nsum, d_bad);
    hipLaunchKernelGGL(K::planes_kernel, dim3((unsigned)((npl + 63) / 64)), dim3(64), 0, s, (const G1Xyzz*)ws.win.ptr, npl, d_bad + 8);
    hipLaunchKernelGGL(K::rowcol_kernel, dim3(2, (unsigned)nj), dim3(msmfb::PLANE_THREADS), msmfb::PLANE_THREADS * sizeof(msmfb::G1Xyzz30), s,
                       (const msmfb::G1Xyzz30*)ws.seg.ptr, d_rc, rs);
    MH_HIP(hipGetLastError());
    char* h = (char*)c.chk.h.ptr;
    MH_HIP(hipMemcpyAsync(h, c.chk.d.ptr, 128 + rc_b, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    const u32* hb = (const u32*)h;
    if (hb[0]) return chk_fail("row / column sums", std::to_string(hb[0]) + " sum(s) off the curve; first: job " + std::to_string(hb[1] / rs.NS) + " sum " + std::to_string(hb[1] % rs.NS));
    if (hb[8]) return chk_fail("planes", std::to_string(hb[8]) + " plane(s) off the curve; first: job " + std::to_string(hb[9] / rs.nplanes) + " plane " + std::to_string(hb[9] % rs.nplanes));
    chk_rowcol.assign(2 * nj, HG1::identity());
    auto xyzz = [](const char* p) {
      HFq X, Y, ZZ, ZZZ;
      memcpy(X.v, p, FQ_B); memcpy(Y.v, p + FQ_B, FQ_B); memcpy(ZZ.v, p + 2 * FQ_B, FQ_B); memcpy(ZZZ.v, p + 3 * FQ_B, FQ_B);
      return HG1::from_xyzz(X, Y, ZZ, ZZZ);
    };
    for (int k = 0; k < nj; k++) {
      const HG1 rows = xyzz(h + 128 + (size_t)(2 * k) * sizeof(G1Xyzz)), cols = xyzz(h + 128 + (size_t)(2 * k + 1) * sizeof(G1Xyzz));
      if (!rows.add(cols.neg()).is_identity()) return chk_fail("row / column sums", "job " + std::to_string(k) + ": the rows and the columns do not sum to the same point");
      chk_rowcol[2 * k] = rows;
    }
    chk_log += "reduce: ok (every sum and plane on the curve; sum of rows = sum of columns)\n";
    c.chk.report = chk_log;
    return MH_OK;
  }
  std::vector<HG1> chk_rowcol;
  // in finish(): the results
  int check_result(hipStream_t s, const HG1* out) {
    if (c.chk.level == 0 || chk_skip) return MH_OK;
    namespace K = msmchk;
    const size_t npl = rs.nplanes;
    const uint64_t* sums_h = (const uint64_t*)ws.h_out.ptr;
    auto on_curve = [](const HG1& p) {
      if (p.is_identity()) return true;
      const HFq z2 = p.Z.sqr(), z6 = z2.sqr() * z2;
      return p.Y.sqr() == p.X.sqr() * p.X + HFq::from_u64(hostff::G1_B) * z6;
    };
    for (int k = 0; k < nj; k++) {
      // the T plane is the sum of all rows
      const uint64_t* p = sums_h + ((size_t)k * npl + (npl - 1)) * XYZZ_L;
      HFq X, Y, ZZ, ZZZ;
      memcpy(X.v, p, FQ_B); memcpy(Y.v, p + FQ_L, FQ_B); memcpy(ZZ.v, p + 2 * FQ_L, FQ_B); memcpy(ZZZ.v, p + 3 * FQ_L, FQ_B);
      if (!HG1::from_xyzz(X, Y, ZZ, ZZZ).add(chk_rowcol[2 * k].neg()).is_identity()) return chk_fail("planes", "job " + std::to_string(k) + ": the T plane is not the sum of the rows");
      if (!on_curve(out[k])) return chk_fail("combine", "job " + std::to_string(k) + ": the result is not on the curve");
    }
    bool full = false;
    if (c.chk.level >= 2 && (u64)nj * nbt <= (1ull << 17)) {
      // the whole reduction again, the classical way, on the host: sum_b (b + 1) B_b over the owned buckets by running sums
      full = true;
      const size_t nbk = (size_t)nj * nbt;
      MH_TRY(c.chk.d.ensure(nbk * sizeof(G1Xyzz)));
      std::vector<uint64_t> hb(nbk * XYZZ_L);
      // (buckets of other ranks' partitions were never written: read as whatever they hold, never used below)
      hipLaunchKernelGGL(K::to_std_kernel, dim3((unsigned)((nbk + 63) / 64)), dim3(64), 0, s, (const msmfb::G1Xyzz30*)ws.buckets.ptr, (G1Xyzz*)c.chk.d.ptr, (u64)nbk);
      MH_HIP(hipGetLastError());
      MH_HIP(hipMemcpyAsync(hb.data(), c.chk.d.ptr, nbk * sizeof(G1Xyzz), hipMemcpyDeviceToHost, s));
      MH_HIP(hipStreamSynchronize(s));
      for (int k = 0; k < nj; k++) {
        HG1 run = HG1::identity(), acc = HG1::identity();
        for (u32 b = nbt; b-- > 0;) {
          const u32 v = b >> pshift;
          if (v % own.stride == own.first) {
            const uint64_t* q = hb.data() + ((size_t)k * nbt + b) * XYZZ_L;
            HFq X, Y, ZZ, ZZZ;
            memcpy(X.v, q, FQ_B); memcpy(Y.v, q + FQ_L, FQ_B); memcpy(ZZ.v, q + 2 * FQ_L, FQ_B); memcpy(ZZZ.v, q + 3 * FQ_L, FQ_B);
            run = run.add(HG1::from_xyzz(X, Y, ZZ, ZZZ));
          }
          acc = acc.add(run);
        }
        if (!acc.add(out[k].neg()).is_identity()) return chk_fail("reduce + combine", "job " + std::to_string(k) + ": the result is not sum (b + 1) B_b of the device's buckets (running sums on the host)");
      }
    }
    chk_log += std::string("result: ok (T plane = sum of rows; results on the curve") + (full ? "; results = running-sum reduction of the buckets on the host" : "") + ")\n";
    c.chk.report = chk_log;
    return MH_OK;
  }
};

// One group of <= MAX_JOBS jobs on the fixed-base path, on the main stream: sort -> accumulate -> reduce -> combine on the host.
// skewed = true (and nothing written) when a bucket is so overfull that the caller should take the variable-base path with its
// pair-tree accumulation instead.  (Rounds 3-4 carried an opt-in two-stream software pipeline over two sub-batches here; it
// measured slower -- 80.8 vs 78.1 ms per proof, profiles/r03a_*: the accumulate kernel fills every SIMD's registers, so a
// second stream only ever ran in its tail -- and is gone; `git show 3c10b42:marlin_amd/csrc/capi.hip` has it.)
static int msm_fb_pipeline(Context& c, const BaseSet& bs, int nj, const size_t* offs, const void* const* d_scalars, const size_t* ns,
                           int is_mont, HG1* out, bool& skewed, const int* shard, bool& partial, const size_t* strides = nullptr,
                           u32 skew_floor = 4096) {
  skewed = false;
  hipStream_t s0 = c.stream;
  ProfScope wall(c, PF_MSM, s0);
  FbRun A(c, bs, c.fbws);
  A.skew_floor = skew_floor;
  for (int k = 0; k < nj; k++) {
    A.offs.push_back(offs[k]); A.sc.push_back(d_scalars[k]); A.ns.push_back(ns[k]); A.strides.push_back(strides ? strides[k] : 1);
  }
  MH_TRY(A.prepare(is_mont, shard));
  partial = A.partial;
  const bool chk = c.chk.level > 0;
  const int hurt = g_debug_corrupt;             // test hook: damage this batch's own data after one stage, once
  g_debug_corrupt = 0;
  MH_TRY(A.sort(s0));
  if (hurt == 1 && A.ent) { const u32 x = 0x00000100u; u32 e = 0; u32* p = (u32*)c.fbws.sorted.ptr + A.ent / 2;
    MH_HIP(hipMemcpyAsync(&e, p, 4, hipMemcpyDeviceToHost, s0)); MH_HIP(hipStreamSynchronize(s0)); e ^= x;
    MH_HIP(hipMemcpyAsync(p, &e, 4, hipMemcpyHostToDevice, s0)); MH_HIP(hipStreamSynchronize(s0)); }
  if (chk) MH_TRY(A.check_sort(s0));
  MH_TRY(A.accum(s0));
  if (hurt == 2) MH_HIP(hipMemsetAsync((msmfb::G1Xyzz30*)c.fbws.buckets.ptr + 5, 0x5A, sizeof(msmfb::G1Xyzz30), s0));
  if (hurt == 3) {            // the heaviest bucket and its neighbour certainly differ
    MH_HIP(hipMemcpyAsync((msmfb::G1Xyzz30*)c.fbws.buckets.ptr + 1, (msmfb::G1Xyzz30*)c.fbws.buckets.ptr, sizeof(msmfb::G1Xyzz30), hipMemcpyDeviceToDevice, s0)); }
  if (chk) MH_TRY(A.check_accum(s0));
  MH_TRY(A.reduce(s0));
  if (hurt == 4) MH_HIP(hipMemcpyAsync((G1Xyzz*)c.fbws.win.ptr + 1, (G1Xyzz*)c.fbws.win.ptr, sizeof(G1Xyzz), hipMemcpyDeviceToDevice, s0));
  if (chk) MH_TRY(A.check_reduce(s0));
  MH_TRY(A.finish(s0, out));
  if (A.skewed) { skewed = true; partial = false; }
  return MH_OK;
}

// Build the window table of a base set: level j = 2^{start_j} * P (c-bit windows tiling 256 bits), affine.
// Automatic window width: lg(n) up to 2^18 points, lg(n) - 1 above, at most 20 (measured on the prover: 2^20-point SRS 51.0 /
// 42.4 / 40.7 / 44.4 ms per proof at c = 16 / 18 / 19 / 20; 2^22-point SRS 115.8 / 113.2 ms at c = 19 / 20).  Multi-GPU proving
// shards every MSM by bucket range, which leaves the width alone.
static uint32_t auto_window_bits(size_t n) {
  static const int env_c = [] { const char* e = getenv("MH_FB_C"); return e ? atoi(e) : 0; }();
  if (env_c) return (uint32_t)env_c;
  size_t eff = n;
  u32 lg = 0;
  while ((1ull << lg) < eff) lg++;
  // Round 4 sweep with the reduction's equal-x doublings fixed (profiles/r04w_window_width_sweep_small_sizes.txt): below 2^19 points
  // the best width is lg(n) itself, not lg(n) - 1 -- at these sizes a bucket thread's chain of additions and the lone wave per
  // SIMD are what costs, and twice the buckets halve the chains (2^14 constraints: 10.5 -> 7.3 ms per proof, 2^12: 8.4 -> 7.4,
  // 2^10: 8.8 -> 6.5); one more bit underloads the H-sized jobs, which then leave for the variable-base path (2^14: 14.7 ms).
  int cc = lg <= 18 ? (int)lg : (int)lg - 1;
  if (cc == 17) cc = 16;               // 17 tiles 256 bits with 16 windows of 16 bits: same digits as 16, twice the buckets
  return (uint32_t)(cc > msmfb::MAX_C ? msmfb::MAX_C : (cc < 8 ? 8 : cc));
}

int bases_precompute(Context& c, BaseSet& bs, uint32_t cbits) {
  if (bs.d_table) { (void)hipFree(bs.d_table); bs.d_table = nullptr; bs.tab_c = bs.tab_W = 0; }
  if (bs.n == 0) return MH_OK;
  bs.tab_auto = cbits == 0;
  if (cbits == 0) cbits = auto_window_bits(bs.n);
  if (cbits < 4 || cbits > (uint32_t)msmfb::MAX_C) return fail(MH_EINVAL, "mh_bases_precompute: window width must be in [4, 20]");
  msm::Windows win;
  const u32 W = msm::make_windows(cbits, win);
  if ((u64)W * bs.n >= (1ull << 31)) return fail(MH_EINVAL, "mh_bases_precompute: table too large for 31-bit entry indices");
  void* tab = nullptr;
  const size_t pt30 = sizeof(msmfb::G1Aff30);
  hipError_t e = hipMalloc(&tab, (size_t)W * bs.n * pt30);
  if (e != hipSuccess) return fail(MH_ENOMEM, "mh_bases_precompute: hipMalloc of the window table failed");
  // two standard-form levels ping-pong through scratch while the chain of doublings runs
  void* tmp = nullptr;                                    // two ping-pong levels in standard form + the un-normalised points of one level
  e = hipMalloc(&tmp, 2 * bs.n * PT_B + bs.n * sizeof(msmfb::G1XyzzStd));
  if (e != hipSuccess) { (void)hipFree(tab); return fail(MH_ENOMEM, "mh_bases_precompute: hipMalloc of the doubling scratch failed"); }
  if (g_debug_poison_scratch) { debug_poison(tab, (size_t)W * bs.n * pt30); debug_poison(tmp, 2 * bs.n * PT_B + bs.n * sizeof(msmfb::G1XyzzStd)); }
  msmfb::G1XyzzStd* xyzz_scratch = (msmfb::G1XyzzStd*)((char*)tmp + 2 * bs.n * PT_B);
  hipStream_t s = c.stream;
  const unsigned grid = (unsigned)(((bs.n + msmfb::TAB_BATCH - 1) / msmfb::TAB_BATCH + 127) / 128);
  const G1Affine* prev = (const G1Affine*)bs.d_points;
  // level 0 = [R^-1 mod r] P (see table_level_kernel): R^-1 as a canonical integer is the value whose Montgomery words are 1
  Fr kinv;
  {
    HFr one_raw = HFr::zero(); one_raw.v[0] = 1;
    uint64_t k64[4];
    one_raw.to_canonical(k64);
    memcpy(kinv.v, k64, 32);
  }
  for (u32 j = 0; j < W; j++) {
    G1Affine* next_std = (G1Affine*)((char*)tmp + (size_t)(j & 1) * bs.n * PT_B);
    hipLaunchKernelGGL(msmfb::table_level_kernel, dim3(grid), dim3(128), 0, s, prev, next_std,
                       (msmfb::G1Aff30*)((char*)tab + (size_t)j * bs.n * pt30), xyzz_scratch, (u64)bs.n, j ? (u32)win.bits[j - 1] : 0u, kinv);
    prev = next_std;
  }
  hipError_t le = hipGetLastError();
  hipError_t se = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  if (le != hipSuccess || se != hipSuccess) { (void)hipFree(tab); return fail(MH_EHIP, "mh_bases_precompute: table kernels failed"); }
  bs.d_table = tab; bs.tab_c = cbits; bs.tab_W = W;
  return MH_OK;
}

// A batch of independent MSMs through one launch sequence.  d_bases[j]: G1Affine[n_j]; d_scalars[j]: Fr[n_j];
// out_xyz: njobs x 18 limbs (Jacobian).  Jobs with n_j == 0 yield the identity.
// shard = {rank, world} (nullptr: one GPU): groups on the fixed-base path are sharded by bucket range and partial[j] = 1
// marks a result that is this rank's share of the sum; every other job is computed in full on every rank (partial[j] = 0).
int msm_batch_device(Context& c, int njobs_in, const void* const* d_bases, const void* const* d_scalars, const size_t* ns,
                     int is_mont, uint64_t* out_xyz, const int* shard, uint8_t* partial_out) {
  if (shard && shard[1] > 1 && !partial_out) return fail(MH_EINVAL, "msm_batch_device: sharded call without a partial-flag array");
  if (partial_out) memset(partial_out, 0, (size_t)njobs_in);
  HG1 id = HG1::identity();
  for (int j = 0; j < njobs_in; j++) { memcpy(out_xyz + XYZ_L * j, id.X.v, FQ_B); memcpy(out_xyz + XYZ_L * j + FQ_L, id.Y.v, FQ_B); memcpy(out_xyz + XYZ_L * j + 2 * FQ_L, id.Z.v, FQ_B); }
  // process in groups of at most MAX_JOBS non-empty jobs
  std::vector<int> live;
  for (int j = 0; j < njobs_in; j++) if (ns[j]) { if (ns[j] >= (1ull << 31)) return fail(MH_EINVAL, "msm: n must be < 2^31"); live.push_back(j); }
  MH_TRY(msm_set_attrs(c));
  hipStream_t s = c.stream;
  // MH_MSM_ALGO = xyzz | tree forces that accumulation on the variable-base path (and disables the fixed-base tables)
  static const int forced = [] { const char* e = getenv("MH_MSM_ALGO"); return !e ? 0 : (std::string(e) == "tree" ? 2 : 1); }();
  size_t g_next = 0;
  for (size_t g0 = 0; g0 < live.size(); g0 = g_next) {
    // a group: up to MAX_JOBS jobs whose entries (windows x scalars) fit 32-bit offsets
    int nj = 0;
    {
      u64 ent_est = 0;
      while (nj < msm::MAX_JOBS && g0 + nj < live.size()) {
        const size_t n = ns[live[g0 + nj]];
        size_t o = 0;
        const BaseSet* tb = forced == 0 ? find_table(c, d_bases[live[g0 + nj]], n, o) : nullptr;
        // a table group may still fall back to the variable-base plan (skew, underload, mixed base sets): size for the larger
        const u64 e = (u64)std::max<u32>(tb ? tb->tab_W : 0, msm::make_plan(n).W) * n;
        if (nj && ent_est + e >= (1ull << 32) - (1ull << 26)) break;
        ent_est += e;
        nj++;
      }
    }
    g_next = g0 + nj;
    size_t nmax = 0, nsum = 0;
    for (int k = 0; k < nj; k++) { size_t n = ns[live[g0 + k]]; nmax = std::max(nmax, n); nsum += n; }
    // fixed-base path: every job of the group lies inside one base set with a window table, and the shared bucket
    // set is reasonably loaded
    if (forced == 0) {
      size_t off0 = 0;
      const BaseSet* bs = find_table(c, d_bases[live[g0]], ns[live[g0]], off0);
      bool ok = bs != nullptr;
      std::vector<size_t> offs(nj); std::vector<const void*> sc(nj); std::vector<size_t> nn(nj);
      for (int k = 0; ok && k < nj; k++) {
        size_t o = 0;
        ok = find_table(c, d_bases[live[g0 + k]], ns[live[g0 + k]], o) == bs;
        offs[k] = o; sc[k] = d_scalars[live[g0 + k]]; nn[k] = ns[live[g0 + k]];
      }
      if (ok && (u64)bs->tab_W * nsum >= 8ull * nj * (1ull << (bs->tab_c - 1))) {
        std::vector<HG1> res(nj);
        bool skewed = false, part = false;
        MH_TRY(msm_fb_pipeline(c, *bs, nj, offs.data(), sc.data(), nn.data(), is_mont, res.data(), skewed, shard, part));
        if (!skewed) {
          c.n_fb_groups++;
          for (int k = 0; k < nj; k++) {
            if (partial_out) partial_out[live[g0 + k]] = part ? 1 : 0;
            uint64_t* o = out_xyz + XYZ_L * live[g0 + k];
            memcpy(o, res[k].X.v, FQ_B); memcpy(o + FQ_L, res[k].Y.v, FQ_B); memcpy(o + 2 * FQ_L, res[k].Z.v, FQ_B);
          }
          continue;
        }
      }
    }
    c.n_vb_groups++;
    msm::Plan p = msm::make_plan(nmax);
    // tiles: aim for ~1024 (window, tile) blocks over the whole batch
    {
      u64 t = (nsum * p.W + 1023) / 1024;
      if (t < 4096) t = 4096;
      if (t > 65536) t = 65536;
      p.tile = (u32)t;
    }
    msm::Jobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    jobs.njobs = nj;
    u64 ent = 0, bho = 0; u32 max_tiles = 0;
    for (int k = 0; k < nj; k++) {
      int j = live[g0 + k];
      jobs.bases[k] = (const G1Affine*)d_bases[j]; jobs.scalars[k] = (const Fr*)d_scalars[j]; jobs.n[k] = ns[j];
      jobs.ent_off[k] = ent; ent += (u64)p.W * ns[j];
      jobs.ntiles[k] = (u32)((ns[j] + p.tile - 1) / p.tile);
      jobs.bh_off[k] = bho; bho += (u64)p.W * jobs.ntiles[k] * p.nb;
      max_tiles = std::max(max_tiles, jobs.ntiles[k]);
    }
    if (ent >= (1ull << 32)) return fail(MH_EINVAL, "msm batch too large for 32-bit entry offsets");
    const u64 WN = ent;
    const u32 WT = p.W * nj;
    const size_t WB = (size_t)WT * p.nb;
    MH_TRY(c.msm_dig.ensure(WN * 4)); MH_TRY(c.msm_sorted.ensure(WN * 4)); MH_TRY(c.msm_bh.ensure(bho * 4));
    MH_TRY(c.msm_tot.ensure(WB * 4)); MH_TRY(c.msm_base.ensure(WB * 4)); MH_TRY(c.msm_pend.ensure(WB * 4));
    MH_TRY(c.msm_buckets.ensure(WB * sizeof(G1Xyzz)));
    MH_TRY(c.msm_seg.ensure((size_t)WT * p.nseg * sizeof(G1Xyzz)));
    MH_TRY(c.msm_win.ensure((size_t)WT * sizeof(G1Xyzz)));
    {
      ProfScope ps(c, PF_MSM);
      hipLaunchKernelGGL(msm::digits_kernel, dim3((unsigned)((nmax + 255) / 256), nj), dim3(256), 0, s, jobs, (u32*)c.msm_dig.ptr,
                         p.W, p.win, is_mont);
      size_t lds = (size_t)p.nb * 4;
      hipLaunchKernelGGL(msm::hist_kernel, dim3(msm::xcd_grid(max_tiles, WT)), dim3(msm::HIST_THREADS), lds, s, jobs,
                         (const u32*)c.msm_dig.ptr, (u32*)c.msm_bh.ptr, p.nb, p.tile, max_tiles, p.W);
      hipLaunchKernelGGL(msm::colscan_kernel, dim3((p.nb + 255) / 256, p.W, nj), dim3(256), 0, s, jobs, (u32*)c.msm_bh.ptr,
                         (u32*)c.msm_tot.ptr, p.nb, p.W);
      MH_TRY(c.tr_sums.ensure(64));
      u32* d_max = (u32*)c.tr_sums.ptr;
      MH_HIP(hipMemsetAsync(d_max, 0, 4, s));
      hipLaunchKernelGGL(msm::binscan_kernel, dim3(WT), dim3(1024), 0, s, (const u32*)c.msm_tot.ptr, (u32*)c.msm_base.ptr, p.nb, d_max, (u32*)nullptr);
      hipLaunchKernelGGL(msm::scatter_kernel, dim3(msm::xcd_grid(max_tiles, WT)), dim3(msm::HIST_THREADS), lds, s, jobs,
                         (const u32*)c.msm_dig.ptr, (const u32*)c.msm_bh.ptr, (const u32*)c.msm_base.ptr, (u32*)c.msm_sorted.ptr,
                         p.nb, p.tile, p.W, max_tiles);
      // accumulate algorithm: XYZZ thread-per-bucket by default; the pair tree when bucket sizes are badly skewed
      // (e.g. many equal scalars), where a thread-per-bucket loop would serialise.  MH_MSM_ALGO=xyzz|tree forces one.
      bool use_tree = forced == 2;
      if (forced == 0) {
        u32 mx = 0;
        MH_HIP(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, s));
        MH_HIP(hipStreamSynchronize(s));
        u64 avg = WN / WB + 1;
        use_tree = mx > 4096 && (u64)mx > 32 * avg;
      }
      if (!use_tree) {
        {
          ProfScope pa(c, PF_MSM_ACCUM);
          // (forcing 2 resident blocks per CU to make the block count an integral number of rounds was measured
          //  and is not faster than letting 3 reside: 15.3 vs 14.6 ms at 2^22)
          const u64 nblk = (WB + msm::ACC_TPB - 1) / msm::ACC_TPB;
          hipLaunchKernelGGL(msm::accum_kernel, dim3((unsigned)nblk), dim3(msm::ACC_TPB), 0, s, jobs, (u32*)c.msm_sorted.ptr,
                             (const u32*)c.msm_base.ptr, (const u32*)c.msm_tot.ptr, (G1Xyzz*)c.msm_buckets.ptr,
                             (u32*)c.msm_pend.ptr, p.nb, p.W, (u64)WB);
        }
        hipLaunchKernelGGL(msm::fixup_kernel, dim3((unsigned)((WB + 63) / 64)), dim3(64), 0, s, jobs, (const u32*)c.msm_sorted.ptr,
                           (const u32*)c.msm_base.ptr, (const u32*)c.msm_pend.ptr, (G1Xyzz*)c.msm_buckets.ptr, p.nb, p.W, (u64)WB);
      } else {
        ProfScope pa(c, PF_MSM_ACCUM);
        MH_TRY(msm_tree_accumulate(c, jobs, WN, p));
      }
      hipLaunchKernelGGL(msm::reduce1_kernel, dim3((WT * p.nseg + 63) / 64), dim3(64), 0, s, (const G1Xyzz*)c.msm_buckets.ptr,
                         (G1Xyzz*)c.msm_seg.ptr, p.nb, p.nseg, WT, (u32)msm::SEG);
      hipLaunchKernelGGL(msm::reduce2_kernel, dim3(1, WT), dim3(256), 0, s, (const G1Xyzz*)c.msm_seg.ptr, (G1Xyzz*)c.msm_win.ptr, p.nseg);
      MH_HIP(hipGetLastError());
    }
    std::vector<uint64_t> win((size_t)WT * XYZZ_L);
    MH_HIP(hipMemcpyAsync(win.data(), c.msm_win.ptr, win.size() * 8, hipMemcpyDeviceToHost, s));
    MH_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < nj; k++) {
      std::vector<uint64_t> wj(win.begin() + (size_t)k * p.W * XYZZ_L, win.begin() + (size_t)(k + 1) * p.W * XYZZ_L);
      HG1 r = combine_windows(wj, p);
      uint64_t* o = out_xyz + XYZ_L * live[g0 + k];
      memcpy(o, r.X.v, FQ_B); memcpy(o + FQ_L, r.Y.v, FQ_B); memcpy(o + 2 * FQ_L, r.Z.v, FQ_B);
    }
  }
  return MH_OK;
}

// MSMs over STRIDED selections of one tabled base set: job j multiplies scalar i with base first[j] + i * stride (the cyclic
// slice of a coefficient vector a rank holds in the distributed layout of ntt_dist.cuh: first = rank + offset, stride = number
// of ranks).  Fixed-base path only -- the window table is indexed, no base is ever copied -- and never the skew fallback.
int msm_batch_strided_device(Context& c, const BaseSet& bs, int njobs, const size_t* first, size_t stride, const void* const* d_scalars,
                             const size_t* ns, int is_mont, uint64_t* out_xyz) {
  if (!bs.d_table) return fail(MH_EINVAL, "strided MSM: the base set has no window table (mh_bases_precompute)");
  if (stride == 0) return fail(MH_EINVAL, "strided MSM: stride must be positive");
  HG1 id = HG1::identity();
  for (int j = 0; j < njobs; j++) { memcpy(out_xyz + XYZ_L * j, id.X.v, FQ_B); memcpy(out_xyz + XYZ_L * j + FQ_L, id.Y.v, FQ_B); memcpy(out_xyz + XYZ_L * j + 2 * FQ_L, id.Z.v, FQ_B); }
  std::vector<int> live;
  for (int j = 0; j < njobs; j++)
    if (ns[j]) {
      if (first[j] + (ns[j] - 1) * stride >= bs.n) return fail(MH_EINVAL, "strided MSM: selection reaches past the base set");
      live.push_back(j);
    }
  MH_TRY(msm_set_attrs(c));
  size_t g_next = 0;
  for (size_t g0 = 0; g0 < live.size(); g0 = g_next) {
    int nj = 0;
    u64 ent_est = 0;
    while (nj < msm::MAX_JOBS && g0 + nj < live.size()) {
      const u64 e = (u64)bs.tab_W * ns[live[g0 + nj]];
      if (nj && ent_est + e >= (1ull << 32) - (1ull << 26)) break;
      ent_est += e; nj++;
    }
    g_next = g0 + nj;
    std::vector<size_t> offs(nj), nn(nj), st(nj, stride); std::vector<const void*> sc(nj);
    for (int k = 0; k < nj; k++) { offs[k] = first[live[g0 + k]]; nn[k] = ns[live[g0 + k]]; sc[k] = d_scalars[live[g0 + k]]; }
    std::vector<HG1> res(nj);
    bool skewed = false, part = false;
    MH_TRY(msm_fb_pipeline(c, bs, nj, offs.data(), sc.data(), nn.data(), is_mont, res.data(), skewed, nullptr, part, st.data()));
    // A skewed batch -- heavily repeated digits, e.g. a round polynomial with one dominant coefficient value -- leaves the
    // fixed-base path like a contiguous one does: the bases of every slice are gathered into a contiguous scratch vector and the
    // group takes the variable-base path, whose pair tree has no long per-thread list.  (Round 5 re-ran the fixed-base pipeline with
    // the per-bucket limit at 2^22 instead: one thread then walked a list of millions for seconds while the peer ranks sat in the
    // next collective -- ADVICE r05.)  A valid sliced proof completes on every rank, as the replicated and the one-GPU prover do.
    if (skewed) {
      size_t tot = 0;
      for (int k = 0; k < nj; k++) tot += nn[k];
      MH_TRY(c.msm_gather.ensure(tot * PT_B));
      std::vector<const void*> gb(nj);
      size_t o = 0;
      for (int k = 0; k < nj; k++) {
        gb[k] = (const char*)c.msm_gather.ptr + o * PT_B;
        hipLaunchKernelGGL(gather_strided_kernel, dim3((unsigned)((nn[k] + 255) / 256)), dim3(256), 0, c.stream, (G1Affine*)c.msm_gather.ptr + o,
                           (const G1Affine*)bs.d_points, (u64)offs[k], (u64)stride, (u64)nn[k]);
        o += nn[k];
      }
      MH_HIP(hipGetLastError());
      std::vector<uint64_t> xyz((size_t)XYZ_L * nj);
      MH_TRY(msm_batch_device(c, nj, gb.data(), sc.data(), nn.data(), is_mont, xyz.data(), nullptr, nullptr));
      for (int k = 0; k < nj; k++) memcpy(out_xyz + XYZ_L * live[g0 + k], xyz.data() + XYZ_L * k, XYZ_L * 8);
      continue;
    }
    c.n_fb_groups++;
    for (int k = 0; k < nj; k++) {
      uint64_t* o = out_xyz + XYZ_L * live[g0 + k];
      memcpy(o, res[k].X.v, FQ_B); memcpy(o + FQ_L, res[k].Y.v, FQ_B); memcpy(o + 2 * FQ_L, res[k].Z.v, FQ_B);
    }
  }
  return MH_OK;
}

int msm_device(Context& c, const void* d_bases, const void* d_scalars, int is_mont, size_t n, uint64_t* out_xyz) {
  const void* b[1] = {d_bases}; const void* sc[1] = {d_scalars}; size_t ns[1] = {n};
  return msm_batch_device(c, 1, b, sc, ns, is_mont, out_xyz, nullptr, nullptr);
}

// --------------------------------------------------------------------------------
// G2: multi-scalar multiplication and fixed-base powers (msm_g2.cuh)
// --------------------------------------------------------------------------------
constexpr size_t G2PT_B = 4 * FQ_B;                 // affine x.c0 | x.c1 | y.c0 | y.c1
// out_words: 4 * Fq::N Montgomery words of the affine result followed by the infinity flag
static int msm_g2_device(Context& c, const G2Affine* d_bases, const Fr* d_scalars, int is_mont, size_t n, u32* out_words) {
  const size_t OW = 4 * Fq::N + 1;
  if (n == 0) { memset(out_words, 0, OW * 4); out_words[OW - 1] = 1; return MH_OK; }
  if (n >= (1ull << 28)) return fail(MH_EINVAL, "mh_g2_msm: n must be < 2^28");
  MH_TRY(msm_set_attrs(c));
  hipStream_t s = c.stream;
  msm::Plan p = msm::make_plan(n);
  msm::Jobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  jobs.njobs = 1; jobs.bases[0] = nullptr; jobs.scalars[0] = d_scalars; jobs.n[0] = n;
  jobs.ent_off[0] = 0; jobs.ntiles[0] = p.ntiles; jobs.bh_off[0] = 0;
  const u64 WN = (u64)p.W * n;
  const u32 WT = p.W;
  const size_t WB = (size_t)WT * p.nb;
  MH_TRY(c.msm_dig.ensure(WN * 4)); MH_TRY(c.msm_sorted.ensure(WN * 4)); MH_TRY(c.msm_bh.ensure((u64)p.W * p.ntiles * p.nb * 4));
  MH_TRY(c.msm_tot.ensure(WB * 4)); MH_TRY(c.msm_base.ensure(WB * 4));
  MH_TRY(c.msm_buckets.ensure(WB * sizeof(G2Xyzz)));
  MH_TRY(c.msm_seg.ensure((size_t)WT * p.nseg * sizeof(G2Xyzz)));
  MH_TRY(c.msm_win.ensure((size_t)WT * sizeof(G2Xyzz) + OW * 4));
  MH_TRY(c.tr_sums.ensure(64));
  ProfScope ps(c, PF_MSM);
  hipLaunchKernelGGL(msm::digits_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, s, jobs, (u32*)c.msm_dig.ptr, p.W, p.win, is_mont);
  const size_t lds = (size_t)p.nb * 4;
  hipLaunchKernelGGL(msm::hist_kernel, dim3(msm::xcd_grid(p.ntiles, WT)), dim3(msm::HIST_THREADS), lds, s, jobs, (const u32*)c.msm_dig.ptr,
                     (u32*)c.msm_bh.ptr, p.nb, p.tile, p.ntiles, p.W);
  hipLaunchKernelGGL(msm::colscan_kernel, dim3((p.nb + 255) / 256, p.W, 1), dim3(256), 0, s, jobs, (u32*)c.msm_bh.ptr, (u32*)c.msm_tot.ptr, p.nb, p.W);
  u32* d_max = (u32*)c.tr_sums.ptr;
  MH_HIP(hipMemsetAsync(d_max, 0, 4, s));
  hipLaunchKernelGGL(msm::binscan_kernel, dim3(WT), dim3(1024), 0, s, (const u32*)c.msm_tot.ptr, (u32*)c.msm_base.ptr, p.nb, d_max, (u32*)nullptr);
  hipLaunchKernelGGL(msm::scatter_kernel, dim3(msm::xcd_grid(p.ntiles, WT)), dim3(msm::HIST_THREADS), lds, s, jobs, (const u32*)c.msm_dig.ptr,
                     (const u32*)c.msm_bh.ptr, (const u32*)c.msm_base.ptr, (u32*)c.msm_sorted.ptr, p.nb, p.tile, p.W, p.ntiles);
  hipLaunchKernelGGL(msmg2::accum_kernel, dim3((unsigned)((WB + 127) / 128)), dim3(128), 0, s, d_bases, (const u32*)c.msm_sorted.ptr, (u64)n,
                     (const u32*)c.msm_base.ptr, (const u32*)c.msm_tot.ptr, (G2Xyzz*)c.msm_buckets.ptr, p.nb, (u64)WB);
  hipLaunchKernelGGL(msmg2::reduce1_kernel, dim3((WT * p.nseg + 63) / 64), dim3(64), 0, s, (const G2Xyzz*)c.msm_buckets.ptr, (G2Xyzz*)c.msm_seg.ptr,
                     p.nb, p.nseg, WT, (u32)msm::SEG);
  hipLaunchKernelGGL(msmg2::reduce2_kernel, dim3(WT), dim3(64), 0, s, (const G2Xyzz*)c.msm_seg.ptr, (G2Xyzz*)c.msm_win.ptr, p.nseg);
  u32* d_out = (u32*)((char*)c.msm_win.ptr + (size_t)WT * sizeof(G2Xyzz));
  hipLaunchKernelGGL(msmg2::combine_kernel, dim3(1), dim3(1), 0, s, (const G2Xyzz*)c.msm_win.ptr, p.W, p.win, d_out);
  MH_HIP(hipGetLastError());
  MH_HIP(hipMemcpyAsync(out_words, d_out, OW * 4, hipMemcpyDeviceToHost, s));
  MH_HIP(hipStreamSynchronize(s));
  return MH_OK;
}

static int g2_validate(Context& c, const void* d_points, size_t n) {
  if (n == 0) return MH_OK;
  MH_TRY(c.tr_sums.ensure(64));
  u32* d_bad = (u32*)c.tr_sums.ptr;
  MH_HIP(hipMemsetAsync(d_bad, 0, 4, c.stream));
  hipLaunchKernelGGL(msmg2::check_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, c.stream, (const G2Affine*)d_points, (u64)n, d_bad);
  MH_HIP(hipGetLastError());
  u32 bad = 0;
  MH_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  if (bad) return fail(MH_EINVAL, "G2 bases: " + std::to_string(bad) + " point(s) are not on the twist (the identity cannot be a base)");
  return MH_OK;
}

// --------------------------------------------------------------------------------
// SRS generation (KZG10::setup's powers, known-tau test SRS)
// --------------------------------------------------------------------------------
static int ensure_srs_table(Context& c) {       // Context::srs_table: G1Affine[32][256]
  if (c.srs_table) return MH_OK;
  // T[w][d] = [d * 256^w] G on the host, batch-normalised
  uint64_t gx[FQ_L], gy[FQ_L];
  hostff::g1_generator_canonical(gx, gy);
  HG1Affine g; g.x = HFq::from_canonical(gx); g.y = HFq::from_canonical(gy); g.inf = false;
  const int NW = srs::WINDOWS, NT = srs::TABLE;
  std::vector<HG1> jac((size_t)NW * NT);
  HG1 base = HG1::from_affine(g);
  for (int w = 0; w < NW; w++) {
    jac[(size_t)w * NT] = HG1::identity();
    HG1 acc = HG1::identity();
    for (int d = 1; d < NT; d++) { acc = acc.add(base); jac[(size_t)w * NT + d] = acc; }
    base = acc.add(base);                      // 256 * base
  }
  // batch inversion of Z (skip identities)
  std::vector<HFq> prod(jac.size());
  HFq run = HFq::one();
  for (size_t i = 0; i < jac.size(); i++) { if (!jac[i].is_identity()) run = run * jac[i].Z; prod[i] = run; }
  HFq inv = run.inv();
  std::vector<uint64_t> host(jac.size() * AFF_L, 0);
  for (size_t i = jac.size(); i-- > 0;) {
    if (jac[i].is_identity()) continue;
    HFq prev = HFq::one();
    for (size_t k = i; k-- > 0;) { if (!jac[k].is_identity()) { prev = prod[k]; break; } }
    HFq zi = inv * prev;
    inv = inv * jac[i].Z;
    HFq zi2 = zi.sqr();
    HFq x = jac[i].X * zi2, y = jac[i].Y * zi2 * zi;
    memcpy(&host[i * AFF_L], x.v, FQ_B); memcpy(&host[i * AFF_L + FQ_L], y.v, FQ_B);
  }
  MH_HIP(hipMalloc(&c.srs_table, host.size() * 8));
  MH_HIP(hipMemcpyAsync(c.srs_table, host.data(), host.size() * 8, hipMemcpyHostToDevice, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

int srs_powers_device(Context& c, const uint64_t* tau_mont, const uint64_t* scale_mont, size_t n, size_t first,
                      void* d_out) {
  MH_TRY(ensure_srs_table(c));
  Fr tau, scale;
  memcpy(tau.v, tau_mont, 32); memcpy(scale.v, scale_mont, 32);
  if (n == 0) return MH_OK;
  hipLaunchKernelGGL(srs::powers_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, c.stream, (G1Affine*)d_out,
                     (const G1Affine*)c.srs_table, tau, scale, (u64)n, (u64)first);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

}  // namespace mh

// =====================================================================================
// extern "C"
// =====================================================================================
#define LOCKED_CTX()                                                 \
  Context& c = ctx();                                                \
  CtxLock _lk(c);                                                    \
  if (!c.inited) return fail(MH_ENOINIT, "mh_init has not been called")

extern "C" {

const char* mh_last_error(void) { return g_err.c_str(); }

static int ctx_init(Context& c, int device_id) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(MH_ENODEV, "no HIP device visible");
  if (device_id < 0 || device_id >= count) return fail(MH_EINVAL, "mh_init: device id out of range");
  MH_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  MH_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    return fail(MH_ENODEV, std::string("device is not gfx950: ") + prop.gcnArchName);
  MH_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  c.own_stream = true;
  MH_HIP(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
  { const char* e = getenv("MH_DIAG"); c.diag = e ? (unsigned)atoi(e) : 0u; }
  for (auto& e : c.side_ev) MH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  c.device = device_id;
  c.num_simds = 4 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
  { const char* e = getenv("MH_CHECK"); if (e) c.chk.level = atoi(e); }
  c.inited = true;
  return MH_OK;
}

// the calling thread's current context: the default context unless mh_ctx_set_current chose another (whose device is fixed)
int mh_init(int device_id) {
  Context& c = ctx();
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (c.inited) {
    if (c.device == device_id) return MH_OK;
    return fail(MH_EINVAL, "mh_init: this context is already initialised on another device (mh_ctx_create makes a context for another GPU)");
  }
  return ctx_init(c, device_id);
}

extern "C" int mh_marlin_release_all(void);
int mh_init_devices(const int* device_ids, int n_devices) {
  if (!device_ids || n_devices != 1)
    return fail(MH_EINVAL, "mh_init_devices: a context drives exactly one GPU (one process per GPU with mh_marlin_set_rccl, or one context per GPU "
                           "in one process: mh_ctx_create + mh_marlin_set_local_group)");
  return mh_init(device_ids[0]);
}

static int ctx_shutdown(Context& c) {
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (!c.inited) return MH_OK;
  if (g_multi_ctx) (void)hipSetDevice(c.device);
  (void)hipStreamSynchronize(c.stream);
  {                                      // prover keys, the native transport: they belong to THIS context, whichever thread shuts it down
    Context* saved = t_ctx;
    t_ctx = &c;
    (void)mh_marlin_release_all();
    t_ctx = saved;
  }
  if (c.tw) (void)hipFree(c.tw);
  if (c.tw30) (void)hipFree(c.tw30);
  if (c.tw30s) (void)hipFree(c.tw30s);
  c.tw = nullptr; c.tw30 = nullptr; c.tw30s = nullptr; c.tw_log = 0;
  c.ntt_tmp[0].release(); c.ntt_tmp[1].release(); c.io.release();
  c.msm_dig.release(); c.msm_sorted.release(); c.msm_bh.release(); c.msm_tot.release(); c.msm_base.release();
  c.msm_buckets.release(); c.msm_seg.release(); c.msm_win.release(); c.msm_pend.release(); c.msm_gather.release();
  for (auto& b : c.tr_off) b.release(); for (auto& b : c.tr_cnt) b.release(); for (auto& b : c.tr_p) b.release();
  c.tr_sums.release(); c.tr_ob.release(); c.tr_pre.release(); c.tr_prod.release(); c.tr_scr.release();
  for (auto& kv : c.bases) { if (kv.second.d_points) (void)hipFree(kv.second.d_points); if (kv.second.d_table) (void)hipFree(kv.second.d_table); }
  c.bases.clear();
  for (auto& kv : c.g2_bases) if (kv.second.d_points) (void)hipFree(kv.second.d_points);
  c.g2_bases.clear();
  c.fbws.release_all();
  c.chk.d.release(); c.chk.h.release();
  for (auto& kv : c.ntt_dist_tabs) kv.second.release();
  c.ntt_dist_tabs.clear();
  c.ntt_dist_buf[0].release(); c.ntt_dist_buf[1].release(); c.sl_send.release(); c.sl_recv.release();
  if (c.stream2) { (void)hipStreamSynchronize(c.stream2); (void)hipStreamDestroy(c.stream2); c.stream2 = nullptr; }
  for (auto& e : c.side_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  c.side_job = nullptr; c.side_ran = false;
  if (c.srs_table) { (void)hipFree(c.srs_table); c.srs_table = nullptr; }
  for (auto& r : c.prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  c.prof.clear();
  for (auto e : c.ev_pool) (void)hipEventDestroy(e);
  c.ev_pool.clear();
  if (c.own_stream && c.stream) (void)hipStreamDestroy(c.stream);
  c.stream = nullptr; c.own_stream = false;
  c.ntt_attr_done = c.msm_attr_done = false;
  c.inited = false; c.device = -1;
  return MH_OK;
}
int mh_shutdown(void) { return ctx_shutdown(ctx()); }

// ---- contexts ---------------------------------------------------------------------------------------------------------------
// The reference proves inside ONE process with rayon threads (/root/reference src/ahp/mod.rs:9-10, benches/bench.rs:1-3).  A
// context per GPU (or several per GPU) lets one host process do the same here: every context has its own streams, workspaces,
// base sets, prover keys and shard configuration; a thread binds itself to one with mh_ctx_set_current and then uses the
// ordinary entry points.  Handles (bases, prover keys, device pointers) belong to the context that made them.
static std::mutex g_ctx_mu;
static std::vector<Context*>& all_ctx() { static std::vector<Context*>* v = new std::vector<Context*>(); return *v; }
int mh_ctx_create(int device_id, mh_ctx_t* ctx_out) {
  if (!ctx_out) return fail(MH_EINVAL, "mh_ctx_create: null output");
  *ctx_out = nullptr;
  Context* c = new Context();
  g_multi_ctx = true;
  int prev = -1;
  (void)hipGetDevice(&prev);
  const int rc = ctx_init(*c, device_id);
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc != MH_OK) { delete c; return rc; }
  { std::lock_guard<std::mutex> lk(g_ctx_mu); all_ctx().push_back(c); }
  *ctx_out = reinterpret_cast<mh_ctx_t>(c);
  return MH_OK;
}
static Context* find_ctx(mh_ctx_t h) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  for (Context* c : all_ctx()) if (reinterpret_cast<mh_ctx_t>(c) == h) return c;
  return nullptr;
}
int mh_ctx_set_current(mh_ctx_t h) {
  if (!h) { t_ctx = nullptr; if (g_multi_ctx && default_ctx().inited) (void)hipSetDevice(default_ctx().device); return MH_OK; }
  Context* c = find_ctx(h);
  if (!c) return fail(MH_EINVAL, "mh_ctx_set_current: unknown context");
  t_ctx = c;
  MH_HIP(hipSetDevice(c->device));
  return MH_OK;
}
mh_ctx_t mh_ctx_get_current(void) { return reinterpret_cast<mh_ctx_t>(t_ctx); }
int mh_ctx_destroy(mh_ctx_t h) {
  Context* c = find_ctx(h);
  if (!c) return fail(MH_EINVAL, "mh_ctx_destroy: unknown context");
  const int rc = ctx_shutdown(*c);
  { std::lock_guard<std::mutex> lk(g_ctx_mu); auto& v = all_ctx(); v.erase(std::find(v.begin(), v.end(), c)); }
  if (t_ctx == c) t_ctx = nullptr;
  delete c;
  return rc;
}

int mh_set_stream(void* hip_stream) {
  LOCKED_CTX();
  MH_HIP(hipStreamSynchronize(c.stream));
  if (hip_stream == nullptr) {
    if (!c.own_stream) { MH_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking)); c.own_stream = true; }
    return MH_OK;
  }
  if (c.own_stream && c.stream) (void)hipStreamDestroy(c.stream);
  c.stream = (hipStream_t)hip_stream;
  c.own_stream = false;
  return MH_OK;
}

int mh_synchronize(void) {
  LOCKED_CTX();
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

int mh_device_info(char* name_out, size_t name_cap, int* cu_count, size_t* hbm_bytes) {
  LOCKED_CTX();
  hipDeviceProp_t prop;
  MH_HIP(hipGetDeviceProperties(&prop, c.device));
  if (name_out && name_cap) { snprintf(name_out, name_cap, "%s (%s)", prop.name, prop.gcnArchName); }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return MH_OK;
}

int mh_curve_info(int* curve_id, int* fr_limbs64, int* fq_limbs64, int* fr_two_adicity) {
  if (curve_id) *curve_id = hostff::CURVE_ID;
  if (fr_limbs64) *fr_limbs64 = HFr::N;
  if (fq_limbs64) *fq_limbs64 = HFq::N;
  if (fr_two_adicity) *fr_two_adicity = (int)hostff::FR_TWO_ADICITY_H;
  return MH_OK;
}

int mh_alloc(size_t bytes, void** dptr_out) {
  LOCKED_CTX();
  if (!dptr_out) return fail(MH_EINVAL, "mh_alloc: null out pointer");
  *dptr_out = nullptr;
  if (bytes == 0) return MH_OK;
  MH_HIP(hipMalloc(dptr_out, bytes));
  if (g_debug_poison_scratch) debug_poison(*dptr_out, bytes);
  return MH_OK;
}
int mh_free(void* dptr) {
  LOCKED_CTX();
  if (!dptr) return MH_OK;
  MH_HIP(hipStreamSynchronize(c.stream));
  MH_HIP(hipFree(dptr));
  return MH_OK;
}
int mh_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  LOCKED_CTX();
  if (bytes == 0) return MH_OK;
  MH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}
int mh_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  LOCKED_CTX();
  if (bytes == 0) return MH_OK;
  MH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}
int mh_memcpy_d2d(void* dst, const void* src, size_t bytes) {
  LOCKED_CTX();
  if (bytes == 0) return MH_OK;
  MH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c.stream));
  return MH_OK;
}
int mh_memset(void* dst, int byte, size_t bytes) {
  LOCKED_CTX();
  if (bytes == 0) return MH_OK;
  MH_HIP(hipMemsetAsync(dst, byte, bytes, c.stream));
  return MH_OK;
}

int mh_ntt_dev(int field, const void* d_in, void* d_out, uint32_t log_n, int inverse) {
  LOCKED_CTX();
  if (field != hostff::CURVE_ID) return fail(MH_EINVAL, "mh_ntt: unsupported field");
  if (!d_in || !d_out) return fail(MH_EINVAL, "mh_ntt_dev: null pointer");
  return ntt_device(c, d_in, d_out, log_n, inverse);
}

int mh_ntt_len(int field, uint64_t* data, size_t in_len, uint32_t log_n, int inverse) {
  LOCKED_CTX();
  if (field != hostff::CURVE_ID) return fail(MH_EINVAL, "mh_ntt: unsupported field");
  if (!data) return fail(MH_EINVAL, "mh_ntt: null pointer");
  if (log_n > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "log_n exceeds the two-adicity of Fr");
  if (in_len > ((size_t)1 << log_n)) return fail(MH_EINVAL, "mh_ntt_len: in_len exceeds the domain");
  size_t bytes = (size_t)32 << log_n;
  MH_TRY(c.io.ensure(bytes));
  // only the caller's coefficients go up; the transform reads the rest of the domain as zero (ntt_device_len)
  if (in_len) MH_HIP(hipMemcpyAsync(c.io.ptr, data, in_len * 32, hipMemcpyHostToDevice, c.stream));
  MH_TRY(ntt_device_len(c, c.io.ptr, in_len, c.io.ptr, log_n, inverse));
  MH_HIP(hipMemcpyAsync(data, c.io.ptr, bytes, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}
int mh_ntt(int field, uint64_t* data, uint32_t log_n, int inverse) {
  if (log_n > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "log_n exceeds the two-adicity of Fr");
  return mh_ntt_len(field, data, (size_t)1 << log_n, log_n, inverse);
}

int mh_ntt_coset_dev(int field, const void* d_in, void* d_out, uint32_t log_n, int inverse) {
  LOCKED_CTX();
  if (field != hostff::CURVE_ID) return fail(MH_EINVAL, "mh_ntt_coset: unsupported field");
  if (!d_in || !d_out) return fail(MH_EINVAL, "mh_ntt_coset_dev: null pointer");
  if (log_n > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "log_n exceeds the two-adicity of Fr");
  return ntt_coset_device(c, d_in, d_out, log_n, inverse);
}

int mh_ntt_coset(int field, uint64_t* data, uint32_t log_n, int inverse) {
  LOCKED_CTX();
  if (field != hostff::CURVE_ID) return fail(MH_EINVAL, "mh_ntt_coset: unsupported field");
  if (!data) return fail(MH_EINVAL, "mh_ntt_coset: null pointer");
  if (log_n > hostff::FR_TWO_ADICITY_H) return fail(MH_EINVAL, "log_n exceeds the two-adicity of Fr");
  size_t bytes = (size_t)32 << log_n;
  MH_TRY(c.io.ensure(bytes));
  MH_HIP(hipMemcpyAsync(c.io.ptr, data, bytes, hipMemcpyHostToDevice, c.stream));
  MH_TRY(ntt_coset_device(c, c.io.ptr, c.io.ptr, log_n, inverse));
  MH_HIP(hipMemcpyAsync(data, c.io.ptr, bytes, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

int mh_bases_upload(int curve, const uint64_t* xy, size_t n, uint64_t* handle_out) {
  LOCKED_CTX();
  if (curve != hostff::CURVE_ID) return fail(MH_EINVAL, "unsupported curve");
  if (!handle_out || (!xy && n)) return fail(MH_EINVAL, "mh_bases_upload: null pointer");
  BaseSet b;
  b.n = n;
  if (n) {
    MH_HIP(hipMalloc(&b.d_points, n * PT_B));
    MH_HIP(hipMemcpyAsync(b.d_points, xy, n * PT_B, hipMemcpyHostToDevice, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));
    int rc = bases_validate(c, b.d_points, n);
    if (rc != MH_OK) { (void)hipFree(b.d_points); return rc; }
  }
  uint64_t h = g_next_handle++;
  c.bases[h] = b;
  *handle_out = h;
  return MH_OK;
}

int mh_bases_upload_serialized(int curve, const uint8_t* bytes, size_t n, int compressed, uint64_t* handle_out) {
  LOCKED_CTX();
  if (curve != hostff::CURVE_ID) return fail(MH_EINVAL, "unsupported curve");
  if (!handle_out || (!bytes && n)) return fail(MH_EINVAL, "mh_bases_upload_serialized: null pointer");
  BaseSet b;
  b.n = n;
  if (n) {
    const size_t item = compressed ? FQ_B : 2 * FQ_B;
    MH_TRY(c.io.ensure(n * item));
    MH_TRY(c.tr_sums.ensure(64));
    u32* d_bad = (u32*)c.tr_sums.ptr;
    MH_HIP(hipMemcpyAsync(c.io.ptr, bytes, n * item, hipMemcpyHostToDevice, c.stream));
    MH_HIP(hipMemsetAsync(d_bad, 0, 4, c.stream));
    MH_HIP(hipMalloc(&b.d_points, n * PT_B));
    hipLaunchKernelGGL(g1_deserialize_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, c.stream, (const uint8_t*)c.io.ptr,
                       (G1Affine*)b.d_points, (u64)n, compressed ? 1 : 0, d_bad);
    u32 bad = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) { (void)hipFree(b.d_points); return fail(MH_EHIP, std::string("mh_bases_upload_serialized: ") + hipGetErrorString(e)); }
    if (bad) {
      (void)hipFree(b.d_points);
      return fail(MH_EINVAL, "mh_bases_upload_serialized: " + std::to_string(bad) + " point(s) do not decode to a finite point of the curve "
                             "(SerializationError::InvalidData: coordinate >= p, x^3 + b not a square, bad flags, or the identity)");
    }
  }
  uint64_t h = g_next_handle++;
  c.bases[h] = b;
  *handle_out = h;
  return MH_OK;
}

int mh_bases_from_dev(int curve, const void* d_xy, size_t n, uint64_t* handle_out) {
  LOCKED_CTX();
  if (curve != hostff::CURVE_ID) return fail(MH_EINVAL, "unsupported curve");
  if (!handle_out || (!d_xy && n)) return fail(MH_EINVAL, "mh_bases_from_dev: null pointer");
  BaseSet b;
  b.n = n;
  if (n) {
    MH_HIP(hipMalloc(&b.d_points, n * PT_B));
    MH_HIP(hipMemcpyAsync(b.d_points, d_xy, n * PT_B, hipMemcpyDeviceToDevice, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));
    int rc = bases_validate(c, b.d_points, n);
    if (rc != MH_OK) { (void)hipFree(b.d_points); return rc; }
  }
  uint64_t h = g_next_handle++;
  c.bases[h] = b;
  *handle_out = h;
  return MH_OK;
}

int mh_srs_powers(int curve, const uint64_t* tau_mont, const uint64_t* scale_mont, size_t first, size_t n,
                  uint64_t* handle_out) {
  LOCKED_CTX();
  if (curve != hostff::CURVE_ID) return fail(MH_EINVAL, "unsupported curve");
  if (!tau_mont || !scale_mont || !handle_out) return fail(MH_EINVAL, "mh_srs_powers: null pointer");
  BaseSet b;
  b.n = n;
  if (n) {
    MH_HIP(hipMalloc(&b.d_points, n * PT_B));
    int rc = srs_powers_device(c, tau_mont, scale_mont, n, first, b.d_points);
    if (rc != MH_OK) { (void)hipFree(b.d_points); return rc; }
    MH_HIP(hipStreamSynchronize(c.stream));
  }
  uint64_t h = g_next_handle++;
  c.bases[h] = b;
  *handle_out = h;
  return MH_OK;
}

int mh_bases_download(uint64_t handle, size_t offset, size_t n, uint64_t* xy_out) {
  LOCKED_CTX();
  auto it = c.bases.find(handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_bases_download: unknown handle");
  if (offset > it->second.n || n > it->second.n - offset) return fail(MH_EINVAL, "mh_bases_download: out of range");
  if (n == 0) return MH_OK;
  MH_HIP(hipMemcpyAsync(xy_out, (const char*)it->second.d_points + offset * PT_B, n * PT_B, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}

int mh_bases_free(uint64_t handle) {
  LOCKED_CTX();
  auto it = c.bases.find(handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_bases_free: unknown handle");
  MH_HIP(hipStreamSynchronize(c.stream));
  if (it->second.d_points) (void)hipFree(it->second.d_points);
  if (it->second.d_table) (void)hipFree(it->second.d_table);
  c.bases.erase(it);
  return MH_OK;
}

int mh_bases_precompute(uint64_t handle, uint32_t window_bits) {
  LOCKED_CTX();
  auto it = c.bases.find(handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_bases_precompute: unknown handle");
  return bases_precompute(c, it->second, window_bits);
}

int mh_bases_table_info(uint64_t handle, uint32_t* window_bits, uint32_t* windows, uint64_t* table_bytes) {
  LOCKED_CTX();
  auto it = c.bases.find(handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_bases_table_info: unknown handle");
  const BaseSet& bs = it->second;
  if (window_bits) *window_bits = bs.d_table ? bs.tab_c : 0;
  if (windows) *windows = bs.d_table ? bs.tab_W : 0;
  if (table_bytes) *table_bytes = bs.d_table ? (uint64_t)bs.tab_W * bs.n * sizeof(msmfb::G1Aff30) : 0;
  return MH_OK;
}

int mh_msm_path_counts(uint64_t* fixed_base_groups, uint64_t* variable_base_groups) {
  LOCKED_CTX();
  if (fixed_base_groups) *fixed_base_groups = c.n_fb_groups;
  if (variable_base_groups) *variable_base_groups = c.n_vb_groups;
  return MH_OK;
}

int mh_bases_len(uint64_t handle, size_t* n_out) {
  LOCKED_CTX();
  auto it = c.bases.find(handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_bases_len: unknown handle");
  if (n_out) *n_out = it->second.n;
  return MH_OK;
}

int mh_msm_dev(uint64_t handle, size_t base_offset, const void* d_scalars, int is_mont, size_t n, uint64_t* out_xyz) {
  LOCKED_CTX();
  if (!out_xyz) return fail(MH_EINVAL, "mh_msm: null output");
  auto it = c.bases.find(handle);
  if (it == c.bases.end()) return fail(MH_EINVAL, "mh_msm: unknown bases handle");
  if (base_offset > it->second.n || n > it->second.n - base_offset)
    return fail(MH_EINVAL, "mh_msm: base_offset + n exceeds the uploaded base set");
  if (n && !d_scalars) return fail(MH_EINVAL, "mh_msm: null scalars");
  return msm_device(c, (const char*)it->second.d_points + base_offset * PT_B, d_scalars, is_mont, n, out_xyz);
}

int mh_msm_batch_dev(size_t njobs, const uint64_t* handles, const size_t* base_offsets, const void* const* d_scalars,
                     const size_t* ns, int is_mont, uint64_t* out_xyz) {
  LOCKED_CTX();
  if (njobs && (!handles || !base_offsets || !d_scalars || !ns || !out_xyz)) return fail(MH_EINVAL, "mh_msm_batch_dev: null pointer");
  std::vector<const void*> b(njobs), sc(njobs);
  for (size_t j = 0; j < njobs; j++) {
    auto it = c.bases.find(handles[j]);
    if (it == c.bases.end()) return fail(MH_EINVAL, "mh_msm_batch_dev: unknown bases handle");
    if (base_offsets[j] > it->second.n || ns[j] > it->second.n - base_offsets[j])
      return fail(MH_EINVAL, "mh_msm_batch_dev: base_offset + n exceeds the uploaded base set");
    if (ns[j] && !d_scalars[j]) return fail(MH_EINVAL, "mh_msm_batch_dev: null scalars");
    b[j] = (const char*)it->second.d_points + base_offsets[j] * PT_B;
    sc[j] = d_scalars[j];
  }
  return msm_batch_device(c, (int)njobs, b.data(), sc.data(), ns, is_mont, out_xyz);
}

// mh_msm_batch_dev for scalars in HOST memory: the polynomials of ONE PC::commit call (a Rust host holds them in Vec<Fr>,
// src/lib.rs:172,193,213) go to the device back to back and run as one batched launch sequence -- one sort / accumulate /
// reduce per call instead of one per polynomial (the seam-route measurement of bench.py: 15 separate mh_msm calls spend
// 73 ms accumulating and 26 ms in latency-bound sort / reduce stages, against 56 + 12 when batched by commit round).
int mh_msm_batch(size_t njobs, const uint64_t* handles, const size_t* base_offsets, const uint64_t* const* scalars, const size_t* ns,
                 int is_mont, uint64_t* out_xyz) {
  LOCKED_CTX();
  if (njobs && (!handles || !base_offsets || !scalars || !ns || !out_xyz)) return fail(MH_EINVAL, "mh_msm_batch: null pointer");
  size_t total = 0;
  for (size_t j = 0; j < njobs; j++) { if (ns[j] && !scalars[j]) return fail(MH_EINVAL, "mh_msm_batch: null scalars"); total += ns[j]; }
  MH_TRY(c.io.ensure(total * 32 + 32));
  std::vector<const void*> b(njobs), sc(njobs);
  size_t off = 0;
  for (size_t j = 0; j < njobs; j++) {
    auto it = c.bases.find(handles[j]);
    if (it == c.bases.end()) return fail(MH_EINVAL, "mh_msm_batch: unknown bases handle");
    if (base_offsets[j] > it->second.n || ns[j] > it->second.n - base_offsets[j])
      return fail(MH_EINVAL, "mh_msm_batch: base_offset + n exceeds the uploaded base set");
    b[j] = (const char*)it->second.d_points + base_offsets[j] * PT_B;
    // a job whose host vector IS an earlier job's (a degree-bounded polynomial committed against powers and shifted powers)
    // is uploaded once and shares the device copy -- which also lets the fixed-base path share its sorted lists
    sc[j] = nullptr;
    for (size_t k = 0; k < j; k++) if (scalars[k] == scalars[j] && ns[k] == ns[j]) { sc[j] = sc[k]; break; }
    if (sc[j]) continue;
    sc[j] = (const char*)c.io.ptr + off * 32;
    if (ns[j]) MH_HIP(hipMemcpyAsync((char*)c.io.ptr + off * 32, scalars[j], ns[j] * 32, hipMemcpyHostToDevice, c.stream));
    off += ns[j];
  }
  return msm_batch_device(c, (int)njobs, b.data(), sc.data(), ns, is_mont, out_xyz);
}

int mh_msm(uint64_t handle, size_t base_offset, const uint64_t* scalars, int is_mont, size_t n, uint64_t* out_xyz) {
  LOCKED_CTX();
  if (n && !scalars) return fail(MH_EINVAL, "mh_msm: null scalars");
  MH_TRY(c.io.ensure(n * 32));
  if (n) MH_HIP(hipMemcpyAsync(c.io.ptr, scalars, n * 32, hipMemcpyHostToDevice, c.stream));
  return mh_msm_dev(handle, base_offset, c.io.ptr, is_mont, n, out_xyz);
}

int mh_g1_to_affine(const uint64_t* xyz, uint64_t* xy_out, int* inf_out) {
  if (!xyz || !xy_out) return fail(MH_EINVAL, "mh_g1_to_affine: null pointer");
  HG1 p;
  memcpy(p.X.v, xyz, FQ_B); memcpy(p.Y.v, xyz + FQ_L, FQ_B); memcpy(p.Z.v, xyz + 2 * FQ_L, FQ_B);
  HG1Affine a = p.to_affine();
  if (a.inf) { a.x = HFq::zero(); a.y = HFq::one(); }   // arkworks GroupAffine::zero() = (0, 1, true)
  memcpy(xy_out, a.x.v, FQ_B); memcpy(xy_out + FQ_L, a.y.v, FQ_B);
  if (inf_out) *inf_out = a.inf ? 1 : 0;
  return MH_OK;
}

// sum of n Jacobian points (host; needs no device): combines the per-GPU partial MSM results that the
// ranks exchange with all_gather (elliptic-curve addition is not an RCCL reduction operator).
int mh_g1_sum(const uint64_t* xyz_points, size_t n, uint64_t* out_xyz) {
  if ((!xyz_points && n) || !out_xyz) return fail(MH_EINVAL, "mh_g1_sum: null pointer");
  HG1 acc = HG1::identity();
  for (size_t i = 0; i < n; i++) {
    HG1 p;
    memcpy(p.X.v, xyz_points + XYZ_L * i, FQ_B); memcpy(p.Y.v, xyz_points + XYZ_L * i + FQ_L, FQ_B); memcpy(p.Z.v, xyz_points + XYZ_L * i + 2 * FQ_L, FQ_B);
    acc = acc.add(p);
  }
  memcpy(out_xyz, acc.X.v, FQ_B); memcpy(out_xyz + FQ_L, acc.Y.v, FQ_B); memcpy(out_xyz + 2 * FQ_L, acc.Z.v, FQ_B);
  return MH_OK;
}

// ---- G2 ------------------------------------------------------------------------------------------------------
int mh_g2_bases_upload(int curve, const uint64_t* xy, size_t n, uint64_t* handle_out) {
  LOCKED_CTX();
  if (curve != hostff::CURVE_ID) return fail(MH_EINVAL, "unsupported curve");
  if (!handle_out || (!xy && n)) return fail(MH_EINVAL, "mh_g2_bases_upload: null pointer");
  G2Set b;
  b.n = n;
  if (n) {
    MH_HIP(hipMalloc(&b.d_points, n * G2PT_B));
    MH_HIP(hipMemcpyAsync(b.d_points, xy, n * G2PT_B, hipMemcpyHostToDevice, c.stream));
    MH_HIP(hipStreamSynchronize(c.stream));
    int rc = g2_validate(c, b.d_points, n);
    if (rc != MH_OK) { (void)hipFree(b.d_points); return rc; }
  }
  uint64_t h = g_next_handle++;
  c.g2_bases[h] = b;
  *handle_out = h;
  return MH_OK;
}
int mh_g2_srs_powers(int curve, const uint64_t* gen_xy, const uint64_t* tau_mont, const uint64_t* scale_mont, size_t first, size_t n,
                     uint64_t* handle_out) {
  LOCKED_CTX();
  if (curve != hostff::CURVE_ID) return fail(MH_EINVAL, "unsupported curve");
  if (!handle_out || !gen_xy || !tau_mont) return fail(MH_EINVAL, "mh_g2_srs_powers: null pointer");
  G2Affine H;
  memcpy(&H, gen_xy, G2PT_B);
  {
    void* d_h = nullptr;
    MH_HIP(hipMalloc(&d_h, G2PT_B));
    MH_HIP(hipMemcpyAsync(d_h, gen_xy, G2PT_B, hipMemcpyHostToDevice, c.stream));
    int rc = g2_validate(c, d_h, 1);
    (void)hipFree(d_h);
    if (rc != MH_OK) return rc;
  }
  Fr tau, scale;
  memcpy(tau.v, tau_mont, 32);
  if (scale_mont) memcpy(scale.v, scale_mont, 32); else { HFr one = HFr::one(); memcpy(scale.v, one.v, 32); }
  G2Set b;
  b.n = n;
  if (n) {
    MH_HIP(hipMalloc(&b.d_points, n * G2PT_B));
    MH_TRY(c.tr_sums.ensure(64));
    u32* d_zero = (u32*)c.tr_sums.ptr;
    MH_HIP(hipMemsetAsync(d_zero, 0, 4, c.stream));
    hipLaunchKernelGGL(msmg2::powers_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, (G2Affine*)b.d_points, H, tau, scale,
                       (u64)first, (u64)n, d_zero);
    hipError_t le = hipGetLastError();
    u32 nz = 0;
    hipError_t ce = hipMemcpyAsync(&nz, d_zero, 4, hipMemcpyDeviceToHost, c.stream);
    hipError_t se = hipStreamSynchronize(c.stream);
    if (le != hipSuccess || ce != hipSuccess || se != hipSuccess) { (void)hipFree(b.d_points); return fail(MH_EHIP, "mh_g2_srs_powers: kernel failed"); }
    if (nz) { (void)hipFree(b.d_points); return fail(MH_EINVAL, "mh_g2_srs_powers: a power is the identity (tau or scale is zero)"); }
  }
  uint64_t h = g_next_handle++;
  c.g2_bases[h] = b;
  *handle_out = h;
  return MH_OK;
}
int mh_g2_bases_download(uint64_t handle, size_t offset, size_t n, uint64_t* xy_out) {
  LOCKED_CTX();
  auto it = c.g2_bases.find(handle);
  if (it == c.g2_bases.end()) return fail(MH_EINVAL, "unknown G2 bases handle");
  if (offset + n > it->second.n) return fail(MH_EINVAL, "mh_g2_bases_download: range exceeds the base set");
  if (n == 0) return MH_OK;
  MH_HIP(hipMemcpyAsync(xy_out, (const char*)it->second.d_points + offset * G2PT_B, n * G2PT_B, hipMemcpyDeviceToHost, c.stream));
  MH_HIP(hipStreamSynchronize(c.stream));
  return MH_OK;
}
int mh_g2_bases_free(uint64_t handle) {
  LOCKED_CTX();
  auto it = c.g2_bases.find(handle);
  if (it == c.g2_bases.end()) return fail(MH_EINVAL, "unknown G2 bases handle");
  if (it->second.d_points) (void)hipFree(it->second.d_points);
  c.g2_bases.erase(it);
  return MH_OK;
}
int mh_g2_msm(uint64_t handle, size_t base_offset, const uint64_t* scalars, int is_mont, size_t n, uint64_t* out_xy_mont, int* is_infinity_out) {
  LOCKED_CTX();
  auto it = c.g2_bases.find(handle);
  if (it == c.g2_bases.end()) return fail(MH_EINVAL, "unknown G2 bases handle");
  if (base_offset + n > it->second.n) return fail(MH_EINVAL, "mh_g2_msm: range exceeds the base set");
  if (!out_xy_mont || (!scalars && n)) return fail(MH_EINVAL, "mh_g2_msm: null pointer");
  MH_TRY(c.io.ensure(n * 32 + 32));
  if (n) MH_HIP(hipMemcpyAsync(c.io.ptr, scalars, n * 32, hipMemcpyHostToDevice, c.stream));
  u32 words[4 * Fq::N + 1];
  MH_TRY(msm_g2_device(c, (const G2Affine*)((const char*)it->second.d_points + base_offset * G2PT_B), (const Fr*)c.io.ptr, is_mont, n, words));
  memcpy(out_xy_mont, words, G2PT_B);
  if (is_infinity_out) *is_infinity_out = (int)words[4 * Fq::N];
  return MH_OK;
}

// MH_CHECK at run time: 0 = off, 1 = the cheap invariants of every fixed-base MSM batch, 2 = plus a second, independent computation
// of every bucket list, every bucket and (small bucket sets) the result (msm_check.cuh).  A violation fails the call with MH_ECHECK.
int mh_check_level(int level) {
  LOCKED_CTX();
  if (level < 0 || level > 2) return fail(MH_EINVAL, "mh_check_level: level must be 0, 1 or 2");
  c.chk.level = level;
  return MH_OK;
}
// text of the last checked batch, stage by stage (NUL-terminated, truncated to cap); counts2 (may be NULL): batches checked, violations
int mh_check_report(char* out, size_t cap, uint64_t* counts2) {
  LOCKED_CTX();
  if (out && cap) snprintf(out, cap, "%s", c.chk.report.c_str());
  if (counts2) { counts2[0] = c.chk.batches; counts2[1] = c.chk.violations; }
  return MH_OK;
}

int mh_prof_enable(int on) {
  LOCKED_CTX();
  c.prof_on = on != 0;
  c.prof_mask = (on == 0 || on == 1) ? ~0u : ((unsigned)on >> 1);
  return MH_OK;
}
static int prof_drain(Context& c) {
  if (c.prof.empty()) return MH_OK;
  MH_HIP(hipStreamSynchronize(c.stream));
  for (auto& r : c.prof) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c.prof_ms[r.family] += ms; c.prof_n[r.family]++; }
    c.ev_pool.push_back(r.a); c.ev_pool.push_back(r.b);
  }
  c.prof.clear();
  return MH_OK;
}
int mh_prof_reset(void) {
  LOCKED_CTX();
  MH_TRY(prof_drain(c));
  for (int i = 0; i < PF_COUNT; i++) { c.prof_ms[i] = 0; c.prof_n[i] = 0; }
  return MH_OK;
}
int mh_prof_get(int family, double* total_ms_out, uint64_t* launches_out) {
  LOCKED_CTX();
  if (family < 0 || family >= PF_COUNT) return fail(MH_EINVAL, "mh_prof_get: bad family");
  MH_TRY(prof_drain(c));
  if (total_ms_out) *total_ms_out = c.prof_ms[family];
  if (launches_out) *launches_out = c.prof_n[family];
  return MH_OK;
}

}  // extern "C"
