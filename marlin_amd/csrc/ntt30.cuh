// NTT passes on 30-bit limbs (the kernel the prover's transforms run on).
//
// Same transform, same pass structure and Stockham indexing as ntt.cuh (which stays as the 32-bit cross-check,
// MH_NTT=32): ceil(log n / 8) out-of-place passes over R x C tiles held in LDS.  What changes is the inside of a pass:
//
//  * Arithmetic.  An Fr multiplication on 8 x 32-bit limbs is 128 v_mad_u64_u32 + 128 v_addc_co_u32 (a 64-bit
//    accumulator overflows after two products) plus conditional corrections in every addition and subtraction.  On
//    9 x 30-bit limbs (R' = 2^270) it is 162 v_mad_u64_u32 with one mask + shift per column (gen_fq30.py, the same
//    generator as the base field of the bucket accumulation), and because R' / r = 2^15 the butterflies never reduce:
//    a' = a + w b, b' = a - w b + 2 r grow by at most 2 r per stage; the passes in between store the lazy 9-limb values as
//    they are (36 B per element in the scratch buffers) and only the last pass reduces to the canonical residue.  The transform is linear, so the DATA keeps its standard representation: an element a R (R = 2^256)
//    is re-sliced into 30-bit limbs as the integer it is, multiplied by twiddles held as w R' mod r, and the Montgomery
//    reduction by R' returns a w R.  Only the twiddle table exists in a second form (tw30, 36 B per entry).
//  * Rounds.  A round takes 2^NS elements per work item into registers, runs NS stages on them and writes them back.
//    Measured on MI355X (profiles/r02c_ntt_variants.txt): NS = 3 (8 elements, 188 VGPRs, 128-thread blocks) is SLOWER than
//    NS = 1 (10.6 / 8.7 ms vs 5.9 ms per proof) -- two waves per SIMD cannot cover the LDS and twiddle latencies of a round
//    -- NS = 2 wins from 2^23 points on (1.20 vs 1.23 ms), NS = 1 below (0.160 vs 0.184 ms at 2^20); LDS padding, four
//    instead of three resident blocks and LDS-staged first-pass twiddles change nothing measurable: the kernel is bound
//    by VALU issue (384 instructions per butterfly at NS = 1, 164 of them v_mad_u64_u32; 519 in ntt.cuh).  An ablation
//    without barriers and without twiddle loads runs 10 % faster, which bounds what further latency hiding can give.
//  * LDS holds the tile limb-major (9 arrays of words), optionally padded (PAD).
//  * Twiddles are read from tw30 through L1/L2: the first pass needs 2^B - 1 per tile (shared by its columns), the second
//    pass's 2.4 MB stay L2-resident, a third pass reads each of its twiddles once from HBM.
//
// Reference semantics: ark-poly 0.3 GeneralEvaluationDomain::{fft, ifft} (call sites /root/reference
// src/ahp/prover.rs:326,350-351,359,365,427,488,532-535,545,655,681).
#pragma once
#include "ff.cuh"
#include "fq30.cuh"

namespace ntt30 {

#ifdef MH_CURVE_BN254
using RP = Fq30Params_BN254_FR;
#define FR30_GEN(fn) fn##_BN254_FR
#else
using RP = Fq30Params_BLS12_381_FR;
#define FR30_GEN(fn) fn##_BLS12_381_FR
#endif

constexpr int NL = 9;
constexpr int MAX_B = 8;
constexpr int MAX_LOGC = 2;
static_assert(RP::NL == NL, "Fr on nine 30-bit limbs");

struct Fr30 { u32 v[NL]; };

// ---- representation changes (no arithmetic: the same integer on a different limb grid) ---------------------------
__device__ __forceinline__ Fr30 slice30(const Fr& x) {
  Fr30 r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = 30 * i, w = bit >> 5, s = bit & 31;
    u32 lo = x.v[w] >> s;
    if (s > 2 && w + 1 < Fr::N) lo |= x.v[w + 1] << (32 - s);
    r.v[i] = lo & M30;
  }
  return r;
}
// value < 2^256, limbs normalised
__device__ __forceinline__ Fr pack32(const Fr30& a) {
  Fr r;
#pragma unroll
  for (int j = 0; j < Fr::N; j++) {
    const int bit = 32 * j, i = bit / 30, s = bit % 30;
    u32 w = a.v[i] >> s;
    if (i + 1 < NL) w |= a.v[i + 1] << (30 - s);
    if (s > 28 && i + 2 < NL) w |= a.v[i + 2] << (60 - s);
    r.v[j] = w;
  }
  return r;
}

// ---- lazily reduced field operations ------------------------------------------------------------------------------
__device__ __forceinline__ Fr30 add30(const Fr30& a, const Fr30& b) {
  Fr30 r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u32 s = a.v[i] + b.v[i] + c;
    if (i < NL - 1) { r.v[i] = s & M30; c = s >> 30; } else r.v[i] = s;
  }
  return r;
}
// a - b + 2 r, for b <= 2 r
__device__ __forceinline__ Fr30 sub30(const Fr30& a, const Fr30& b) {
  Fr30 r;
  int c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    int s = (int)(a.v[i] - b.v[i]) + (int)RP::P2[i] + c;
    if (i < NL - 1) { r.v[i] = (u32)s & M30; c = s >> 30; } else r.v[i] = (u32)s;
  }
  return r;
}
__device__ __forceinline__ Fr30 mul30(const Fr30& a, const u32* w) {
  Fr30 r;
  FR30_GEN(f30_mulredc)(r.v, a.v, w);
  return r;
}
// a -= K r if a >= K r
template <int K>
__device__ __forceinline__ void cond_sub(Fr30& a) {
  u32 t[NL];
  int c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 kp = K == 1 ? RP::P[i] : (K == 2 ? RP::P2[i] : (K == 4 ? RP::P4[i] : (K == 8 ? RP::P8[i] : (K == 16 ? RP::P16[i] : (K == 32 ? RP::P32[i] : RP::P64[i])))));
    int s = (int)(a.v[i] - kp) + c;
    if (i < NL - 1) { t[i] = (u32)s & M30; c = s >> 30; } else { t[i] = (u32)s; c = s; }
  }
  if (c >= 0) {
#pragma unroll
    for (int i = 0; i < NL; i++) a.v[i] = t[i];
  }
}
// a < 128 r  ->  the canonical a < r.  A transform of 2^k points leaves values below (1 + 2 k) r <= 65 r at the end of its
// last pass (every stage adds at most 2 r, and the passes in between store the lazy values as they are).  The quotient
// is estimated from the top limb (bits 240 and up) against r's top 15 bits, rounded so that it never overshoots and
// undershoots by at most 2: one multiple of r is subtracted, two conditional subtractions finish (a ladder of seven
// conditional subtractions costs 2.4 x as many instructions).
__device__ __forceinline__ void reduce_full(Fr30& a) {
  constexpr u32 R_TOP = RP::P[NL - 1];                                   // r >> 240
  constexpr u32 MAGIC = (u32)(0xffffffffull / (R_TOP + 1));              // <= 2^32 / (r_top + 1)
  const u32 q = (u32)(((u64)a.v[NL - 1] * MAGIC) >> 32);                 // <= floor(a / r), >= floor(a / r) - 2
  long long c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const long long s = (long long)a.v[i] - (long long)((u64)q * RP::P[i]) + c;
    if (i < NL - 1) { a.v[i] = (u32)s & M30; c = s >> 30; } else a.v[i] = (u32)s;
  }
  cond_sub<2>(a); cond_sub<1>(a);
}
// (a, b) <- (a + w b, a - w b + 2 r): w b < 1.01 r whatever the (lazily reduced) b, so both grow by at most 2 r
__device__ __forceinline__ void butterfly(Fr30& a, Fr30& b, const u32* w) {
  const Fr30 t = mul30(b, w);
  b = sub30(a, t);
  a = add30(a, t);
}

// ---- the same butterfly with the twiddle product as a Shoup multiplication (the default of the one-stage rounds) -----------------------
// The twiddle is a constant: with w' = floor(w R' / r) stored beside the PLAIN w (18 words per entry, tw30s), w b mod r is
// t = w b - q r, q ~ floor(b w' / R') -- 135 limb products (the high half of b w', the low halves of b w and q (R' - r)) instead
// of the Montgomery product's 162, no m_k chain, 18 column carries instead of 27 (gen_fq30.py: shoup).  No Montgomery factor
// either: the data keeps its standard representation as before.  q is taken from the columns i + j >= 8 only, so
// 0 <= t < 11 r; the butterfly subtracts with 16 r: values grow by at most 16 r per stage, (1 + 16 * 28) r < 2^9 r << R' / r.
__device__ __forceinline__ Fr30 sub30_16(const Fr30& a, const Fr30& b) {
  Fr30 r;
  int c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    int s = (int)(a.v[i] - b.v[i]) + (int)RP::P16[i] + c;
    if (i < NL - 1) { r.v[i] = (u32)s & M30; c = s >> 30; } else r.v[i] = (u32)s;
  }
  return r;
}
__device__ __forceinline__ void butterfly_shoup(Fr30& a, Fr30& b, const u32* w) {      // w[0..9) = w, w[9..18) = w'
  Fr30 t;
  FR30_GEN(f30_mulshoup)(t.v, b.v, w, w + NL);
  b = sub30_16(a, t);
  a = add30(a, t);
}
template <bool SHOUP>
__device__ __forceinline__ void butterfly_t(Fr30& a, Fr30& b, const u32* w) {
  if (SHOUP) butterfly_shoup(a, b, w); else butterfly(a, b, w);
}

// ---- tile in LDS: limb-major, 4 words of padding per 32 -------------------------------------------------------------
// PAD: log2 of the words of padding inserted after every 32 words of a limb array (-1: none)
template <int PAD>
__device__ __host__ __forceinline__ u32 phys(u32 a) { return PAD < 0 ? a : a + ((a >> 5) << (PAD < 0 ? 0 : PAD)); }
template <int PAD>
__device__ __forceinline__ Fr30 tile_load(const u32* tile, u32 tw, u32 a) {
  Fr30 r;
  const u32 p = phys<PAD>(a);
#pragma unroll
  for (int l = 0; l < NL; l++) r.v[l] = tile[l * tw + p];
  return r;
}
template <int PAD>
__device__ __forceinline__ void tile_store(u32* tile, u32 tw, u32 a, const Fr30& x) {
  const u32 p = phys<PAD>(a);
#pragma unroll
  for (int l = 0; l < NL; l++) tile[l * tw + p] = x.v[l];
}

// twiddle table in the 30-bit form: tw30[9 i + l] = limb l of (tw[i] R' / R mod r), same indexing as ntt.cuh's table
__global__ __launch_bounds__(256) void build_twiddles30(u32* __restrict__ tw30, const Fr* __restrict__ tw, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr c;
#pragma unroll
  for (int k = 0; k < Fr::N; k++) c.v[k] = RP::TO30[k];
  const Fr30 w = slice30(ff_mul(ff_load(tw + i), c));      // w R * (R' mod r) / R = w R' mod r, canonical
#pragma unroll
  for (int l = 0; l < NL; l++) tw30[9 * i + l] = w.v[l];
}

// the Shoup form of the table: tw30s[18 i + l] = limb l of the plain w_i, tw30s[18 i + 9 + l] = limb l of floor(w_i R' / r).
// With rho = w R' mod r (the entry of tw30): w R' - rho is divisible by r and the quotient is below R', so it is
// (R' - rho) r^-1 mod R' -- the low nine limbs of one product with a constant.
__global__ __launch_bounds__(256) void build_twiddles30s(u32* __restrict__ tw30s, const Fr* __restrict__ tw, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr c, one;
#pragma unroll
  for (int k = 0; k < Fr::N; k++) { c.v[k] = RP::TO30[k]; one.v[k] = k == 0 ? 1u : 0u; }
  const Fr wm = ff_load(tw + i);
  const Fr30 w = slice30(ff_mul(wm, one));                   // w R / R = w, canonical
  const Fr30 rho = slice30(ff_mul(wm, c));                   // w R' mod r, canonical and non-zero (w != 0)
  u32 d[NL];                                                 // R' - rho
  int br = 0;
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const int sdig = -(int)rho.v[l] + br;
    d[l] = (u32)sdig & M30;
    br = sdig >> 30;
  }
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    u64 lo = acc & M30, hi = acc >> 30;                      // two accumulators: nine 60-bit products overflow one
    for (int a = 0; a <= k; a++) {
      const u64 pr = (u64)d[a] * RP::PINV_FULL[k - a];
      lo += pr & M30; hi += pr >> 30;
    }
    hi += lo >> 30;
    tw30s[18 * i + NL + k] = (u32)lo & M30;
    acc = hi;
  }
#pragma unroll
  for (int l = 0; l < NL; l++) tw30s[18 * i + l] = w.v[l];
}

// NS stages (t .. t + NS - 1) of the tile's B on 2^NS elements per work item.  Element j of an item is row
// u0 + j m (m = 2^t) of column c; stage t + s pairs j with j + 2^s (bit s of j clear) under the twiddle of index
// i + (j mod 2^s) m at that stage.  TWL: twiddles from the LDS copy of tw30[1 .. R) (first pass) or from global memory.
template <int NS, int LOGC, bool TWL, int THREADS, int PAD, bool SHOUP = false>
__device__ __forceinline__ void round_stages(u32* tile, u32 tw_words, const u32* __restrict__ twsrc, u32 B, u32 t, u32 logP,
                                             u64 jbase) {
  constexpr int C = 1 << LOGC;
  constexpr int E = 1 << NS;
  const u32 R = 1u << B, m = 1u << t;
  const u32 items = (R >> NS) << LOGC;
  const u64 P = 1ull << logP;
  for (u32 idx = threadIdx.x; idx < items; idx += THREADS) {
    const u32 c = idx & (C - 1), q = idx >> LOGC;
    const u32 i = q & (m - 1), u0 = ((q >> t) << (t + NS)) + i;
    const u64 k = (jbase + c) & (P - 1);
    Fr30 e[E];
#pragma unroll
    for (int j = 0; j < E; j++) e[j] = tile_load<PAD>(tile, tw_words, ((u0 + j * m) << LOGC) + c);
#pragma unroll
    for (int s = 0; s < NS; s++) {
#pragma unroll
      for (int g = 0; g < (1 << s); g++) {            // g = j mod 2^s: one twiddle for all pairs of this residue
        const u64 ti = (P << (t + s)) + k + ((u64)(i + g * m) << logP);
        constexpr int TWW = SHOUP ? 2 * NL : NL;         // words per twiddle
        u32 w[TWW];
        if (TWL) {
#pragma unroll
          for (int l = 0; l < TWW; l++) w[l] = twsrc[TWW * (u32)ti + l];
        } else {
          const u32* p = twsrc + TWW * ti;
#pragma unroll
          for (int l = 0; l < TWW; l++) w[l] = p[l];
        }
#pragma unroll
        for (int j = g; j < E; j += (2 << s)) butterfly_t<SHOUP>(e[j], e[j + (1 << s)], w);
      }
    }
#pragma unroll
    for (int j = 0; j < E; j++) tile_store<PAD>(tile, tw_words, ((u0 + j * m) << LOGC) + c, e[j]);
  }
}

// One Stockham pass (arguments as ntt::pass_kernel; tw30 instead of tw).
template <int LOGC, int MAXNS, int THREADS, int WAVES, int PAD, bool TWLDS, bool SHOUP = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void pass30_kernel(const void* __restrict__ x, void* __restrict__ y, const u32* __restrict__ tw30,
                                                         u32 log_n, u32 B, u32 logP, u32 flags, Fr ninv, u64 in_len) {
  extern __shared__ __attribute__((aligned(16))) u32 lds30[];
  static_assert(!(TWLDS && SHOUP), "the LDS copy of the first pass's twiddles holds the 9-word form");
  constexpr int C = 1 << LOGC;
  const u32 R = 1u << B;
  const u64 n = 1ull << log_n;
  const u64 stride = n >> B;
  const u64 jbase = (u64)blockIdx.x << LOGC;
  const u32 tid = threadIdx.x;
  const u32 tw_words = phys<PAD>(R << LOGC);
  u32* tile = lds30;
  u32* twl = lds30 + NL * tw_words;                  // first pass only: tw30[0 .. R)

  // ---- load, with stage 0 of the pass fused in.  Row r (stride n / R) of column c belongs in tile row bitrev_B(r), and stage 0
  // pairs the tile rows 2u, 2u + 1 = the loaded rows r, r + R / 2 (r < R / 2): one work item loads both, runs the butterfly
  // and stores the results -- one LDS round trip and one barrier less per pass.  Its twiddle is tw[P + k] (k = column mod P);
  // in the first pass (P = 1) that is omega_2^0 = 1 for every butterfly of the stage and the multiplication is skipped:
  // the inputs are reduced field elements there, so a + b < 2 r and a - b + 2 r < 3 r keep the bounds of the lazy
  // arithmetic.  A missing tail (in_len) reads as zero.
  auto load_elem = [&](u64 gi) -> Fr30 {
    Fr30 v;
    if (flags & 4u) {                                  // a pass in between: 9 lazy limbs per element, as the previous pass left them
      const u32* p = (const u32*)x + 9 * gi;
#pragma unroll
      for (int l = 0; l < NL; l++) v.v[l] = p[l];
    } else if (gi < in_len) v = slice30(ff_load((const Fr*)x + gi));
    else {
#pragma unroll
      for (int l = 0; l < NL; l++) v.v[l] = 0;
    }
    return v;
  };
  if (logP == 0 && TWLDS)
    for (u32 idx = tid; idx < R * NL; idx += THREADS) twl[idx] = tw30[idx];
  {
    const u32 half = R >> 1;
    const u64 Pm = (1ull << logP) - 1;
    for (u32 idx = tid; idx < (half << LOGC); idx += THREADS) {
      const u32 c = idx & (C - 1), r = idx >> LOGC;
      const u64 g0 = (jbase + c) + (u64)r * stride;
      Fr30 a = load_elem(g0), b = load_elem(g0 + (u64)half * stride);
      if (logP == 0) {
        const Fr30 d = sub30(a, b);
        a = add30(a, b);
        b = d;
      } else {
        constexpr int TWW = SHOUP ? 2 * NL : NL;
        const u32* w = tw30 + TWW * ((1ull << logP) + ((jbase + c) & Pm));
        u32 wl[TWW];
#pragma unroll
        for (int l = 0; l < TWW; l++) wl[l] = w[l];
        butterfly_t<SHOUP>(a, b, wl);
      }
      const u32 rr = __brev(r) >> (32 - B);            // even: r < R / 2
      tile_store<PAD>(tile, tw_words, (rr << LOGC) + c, a);
      tile_store<PAD>(tile, tw_words, ((rr + 1) << LOGC) + c, b);
    }
  }
  __syncthreads();

  // ---- stages 1 .. B - 1 in rounds of MAXNS (then fewer)
  for (u32 t = 1; t < B;) {
    const u32 ns = B - t >= (u32)MAXNS ? (u32)MAXNS : B - t;
    if (logP == 0 && TWLDS) {
      if (MAXNS >= 3 && ns == 3) round_stages<MAXNS >= 3 ? 3 : 1, LOGC, true, THREADS, PAD>(tile, tw_words, twl, B, t, 0, jbase);
      else if (ns == 2) round_stages<2, LOGC, true, THREADS, PAD>(tile, tw_words, twl, B, t, 0, jbase);
      else round_stages<1, LOGC, true, THREADS, PAD>(tile, tw_words, twl, B, t, 0, jbase);
    } else {
      if (MAXNS >= 3 && ns == 3) round_stages<MAXNS >= 3 ? 3 : 1, LOGC, false, THREADS, PAD, SHOUP>(tile, tw_words, tw30, B, t, logP, jbase);
      else if (ns == 2) round_stages<2, LOGC, false, THREADS, PAD, SHOUP>(tile, tw_words, tw30, B, t, logP, jbase);
      else round_stages<1, LOGC, false, THREADS, PAD, SHOUP>(tile, tw_words, tw30, B, t, logP, jbase);
    }
    __syncthreads();
    t += ns;
  }

  // ---- store.  More passes follow (flags bit 3): the 9 lazy limbs go out as they are, 36 B per element -- no reduction,
  // no repacking; the NTT is bound by VALU issue, not by HBM, so 12 % more bytes on the intermediate buffers are cheaper
  // than ~250 instructions per element.  Last pass: the canonical residue in the standard 8 x 32-bit form, through the
  // multiplication by n^-1 for an inverse transform (index negated).
  const bool inv_last = flags & 1u;
  u32 ninv30[NL];
  if (inv_last) {
    Fr c;
#pragma unroll
    for (int k = 0; k < Fr::N; k++) c.v[k] = RP::TO30[k];
    const Fr30 w = slice30(ff_mul(ninv, c));
#pragma unroll
    for (int l = 0; l < NL; l++) ninv30[l] = w.v[l];
  }
  const u64 P = 1ull << logP;
  for (u32 idx = tid; idx < (R << LOGC); idx += THREADS) {
    const u32 c = idx & (C - 1), r2 = idx >> LOGC;
    Fr30 v = tile_load<PAD>(tile, tw_words, (r2 << LOGC) + c);
    const u64 j = jbase + c;
    const u64 k = j & (P - 1);
    u64 o = ((j - k) << B) + k + ((u64)r2 << logP);        // logP = 0: (j << B) + r2
    if (flags & 8u) {
      u32* q = (u32*)y + 9 * o;
#pragma unroll
      for (int l = 0; l < NL; l++) q[l] = v.v[l];
      continue;
    }
    if (inv_last) {
      o = (n - o) & (n - 1);
      v = mul30(v, ninv30);
      cond_sub<1>(v);
    } else {
      reduce_full(v);
    }
    ff_store((Fr*)y + o, pack32(v));
  }
}

inline size_t pass_lds_bytes(u32 B, int logc, bool first, int pad, bool twlds) {
  const u32 words = (1u << B) << logc;
  const u32 tw_words = pad < 0 ? words : words + ((words >> 5) << pad);
  return ((size_t)NL * tw_words + (first && twlds ? (size_t)NL << B : 0)) * 4;
}

}  // namespace ntt30
