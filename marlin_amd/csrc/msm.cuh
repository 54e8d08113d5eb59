// Pippenger multi-scalar multiplication over BLS12-381 G1 for gfx950.
//
// Computes what the reference obtains from ark-ec 0.3
// `VariableBaseMSM::multi_scalar_mul(bases, scalars)` -- reached only through
// ark-poly-commit's KZG10 commit/open, which the reference calls at
// /root/reference src/lib.rs:125,172,193,213 (PC::commit) and src/lib.rs:292
// (PC::open_combinations); SURVEY.md §8 a7-a9 / Appendix A.  The result is a
// unique group element, so window size, digit signedness and bucket coordinates
// are free design choices; they are chosen for CDNA4, not
// copied from arkworks (which uses unsigned c-bit windows, Jacobian buckets and
// one rayon task per window).
//
// Pipeline (all on one stream; a batch of up to MAX_JOBS independent MSMs goes through every stage together;
// the only host round-trips are a 4-byte read of the largest bucket size and the final copy of the window sums):
//   1. digits   : Montgomery -> canonical scalar, signed base-2^c recoding;
//                 one u32 entry (bucket+1 | sign<<31, 0 = skip) per (window, scalar)
//   2. hist     : per (window, tile) LDS-privatised histogram of bucket ids
//   3. colscan  : per (window, bucket) exclusive prefix over tiles + bucket totals
//   4. binscan  : per window exclusive scan over buckets -> bucket start offsets (+ the largest bucket size)
//   5. scatter  : per (window, tile) LDS-atomic local rank -> point-index lists
//                 grouped by bucket (counting sort; no global atomics anywhere)
//   6. accum    : one thread per (window, bucket): XYZZ += affine base (8M+2S); rare collisions
//                 (base == +-accumulator) deferred to fixup
//   7. reduce1  : per (window, segment of S buckets): running-sum trick inside
//                 the segment + (segment offset) * (segment total) by double-and-add
//   8. reduce2  : per window: tree-sum of the segment results
// The host combines the W window sums (Horner with c doublings each) and
// normalises to affine (host_ff.h).
#pragma once
#include "g1.cuh"

namespace msm {

constexpr int HIST_THREADS = 1024;
constexpr int SEG = 32;  // buckets per reduce1 segment

constexpr int MAX_WINDOWS = 64;

// Window layout: W windows that tile exactly 256 bits (255 scalar bits + 1 bit of
// headroom for the signed-digit carry).  Windows are c or c-1 bits wide, so no
// window is left nearly empty (a 1-bit top window would put every carry into a
// single bucket and serialise its accumulation).
struct Windows {
  unsigned char start[MAX_WINDOWS];   // first bit of window w
  unsigned char bits[MAX_WINDOWS];    // width of window w (c or c-1)
};

struct Plan {
  u32 c;        // widest window, bits
  u32 W;        // number of windows
  u32 nb;       // buckets per window = 2^(c-1)
  u32 tile;     // entries per hist/scatter tile
  u32 ntiles;   // tiles per window
  u32 nseg;     // segments per window
  Windows win;
};

// A batch of MSMs that share one window plan and run through every stage together (the MSMs of one
// PC::commit call are independent: src/lib.rs:172,193,213; batching them fills the GPU during the
// latency-bound sort and reduction stages).  Window index space: gw = job * W + w.
constexpr int MAX_JOBS = 8;
struct Jobs {
  const G1Affine* bases[MAX_JOBS];
  const Fr* scalars[MAX_JOBS];
  u64 n[MAX_JOBS];
  u64 ent_off[MAX_JOBS];   // first element of the job's region in dig / sorted (W * n[j] elements each)
  u64 bh_off[MAX_JOBS];    // first element of the job's region in the per-tile histogram array
  u32 ntiles[MAX_JOBS];
  u32 njobs;
};

// windows of c or c-1 bits (the narrow ones on top) tiling exactly 256 bits; returns their number
inline u32 make_windows(u32 c, Windows& win) {
  const u32 W = (256 + c - 1) / c;
  const u32 narrow = W * c - 256;
  u32 bit = 0;
  for (u32 w = 0; w < W; w++) {
    u32 wb = (w >= W - narrow) ? c - 1 : c;
    win.start[w] = (unsigned char)bit;
    win.bits[w] = (unsigned char)wb;
    bit += wb;
  }
  return W;
}

inline Plan make_plan(u64 n) {
  Plan p;
  u32 lg = 0;
  while ((1ull << lg) < n) lg++;
  // average bucket load ~2^(lg-(c-1)); keep it >= ~64 for large n, LDS histogram caps c at 16
  int c = (int)lg - 5;
  if (c > 16) c = 16;
  if (c < 4) c = 4;
  p.c = (u32)c;
  p.nb = 1u << (c - 1);
  p.W = make_windows((u32)c, p.win);
  u64 t = (n * p.W + 1023) / 1024;       // aim for ~1024 tiles in total
  if (t < 4096) t = 4096;
  if (t > 65536) t = 65536;
  p.tile = (u32)t;
  p.ntiles = (u32)((n + p.tile - 1) / p.tile);
  p.nseg = (p.nb + SEG - 1) / SEG;
  return p;
}

// ---- 1. digits -----------------------------------------------------------------
// signed base-2^c recoding of a canonical scalar: f(w, e) with e = bucket | sign << 31 (bucket in [0, 2^(wb-1)], 0 = skip)
template <class F>
__device__ __forceinline__ void for_each_digit(const Fr& s, u32 W, const Windows& win, F f) {
  u32 carry = 0;
  for (u32 w = 0; w < W; w++) {
    const u32 bit = win.start[w], wb = win.bits[w];
    const u32 half = 1u << (wb - 1);
    const u32 mask = (1u << wb) - 1;
    u32 limb = bit >> 5, sh = bit & 31;
    u64 two = s.v[limb];
    if (limb + 1 < 8) two |= (u64)s.v[limb + 1] << 32;
    u32 raw = ((u32)(two >> sh) & mask) + carry;
    u32 e;
    if (raw > half) { e = ((1u << wb) - raw) | 0x80000000u; carry = 1; }
    else { e = raw; carry = 0; }           // raw == 0 -> skip entry
    f(w, e);
  }
}

// The same recoding for a window width known at compile time (C = the table's c): unrolled, every window's first bit and width are
// constants, so the scalar's words are addressed as registers.  The generic form above indexes s.v[] with a run-time word number,
// which the compiler can only serve from scratch memory: 2 x 13 scratch loads per scalar and pass in the partition kernel.
template <int C, class F>
__device__ __forceinline__ void for_each_digit_c(const Fr& s, F f) {
  constexpr u32 W = (256 + C - 1) / C, narrow = W * C - 256;
  u32 carry = 0;
#pragma unroll
  for (u32 w = 0; w < W; w++) {
    const u32 wb = (w >= W - narrow) ? C - 1 : C;
    const u32 bit = w * C - (w > W - narrow ? w - (W - narrow) : 0);      // = make_windows' start[w]
    const u32 half = 1u << (wb - 1), mask = (1u << wb) - 1;
    const u32 limb = bit >> 5, sh = bit & 31;
    u64 two = s.v[limb];
    if (limb + 1 < 8) two |= (u64)s.v[limb + 1] << 32;
    const u32 raw = ((u32)(two >> sh) & mask) + carry;
    u32 e;
    if (raw > half) { e = ((1u << wb) - raw) | 0x80000000u; carry = 1; }
    else { e = raw; carry = 0; }
    f(w, e);
  }
}

__global__ __launch_bounds__(256) void digits_kernel(Jobs jobs, u32* __restrict__ dig_all, u32 W, Windows win, int is_mont) {
  const u32 job = blockIdx.y;
  const u64 n = jobs.n[job];
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32* dig = dig_all + jobs.ent_off[job];
  Fr s = ff_load(jobs.scalars[job] + i);
  if (is_mont) s = ff_from_mont(s);
  for_each_digit(s, W, win, [&](u32 w, u32 e) { dig[(u64)w * n + i] = e; });
}

// ---- 2. hist ---------------------------------------------------------------------
// grid (ntiles, W); LDS: nb u32 counters
// XCD-aware block -> (window, tile) mapping for the sort kernels.  Workgroup b is observed to run on XCD b % 8
// (MI355X_MICROARCH.md, dispatch; used for speed only).  All tiles of one global window gw = job * W + w are given
// block ids of the same residue mod 8 and consecutive quotients, so each XCD works through "its" windows one
// after another: the scatter's 4-byte writes into that window's 2^(c-1) bucket lists (<= 2 MB of open cache
// lines) then combine in that XCD's 4 MB L2 instead of being written back line by line.
__device__ __forceinline__ void xcd_decode(u32 bid, u32 max_tiles, u32 W, u32 njobs, u32& w, u32& tb, u32& job) {
  const u32 xcd = bid & 7u, k = bid >> 3;
  const u32 slot = k / max_tiles;            // which of this XCD's windows
  tb = k % max_tiles;
  const u32 gw = slot * 8u + xcd;
  job = gw / W;
  w = gw % W;
  (void)njobs;
}
inline u32 xcd_grid(u32 max_tiles, u32 WT) { return ((WT + 7) / 8) * max_tiles * 8; }

__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(Jobs jobs, const u32* __restrict__ dig_all, u32* __restrict__ bh_all,
                                                            u32 nb, u32 tile, u32 max_tiles, u32 W) {
  extern __shared__ __attribute__((aligned(16))) u32 h[];
  u32 w, tb, job;
  xcd_decode(blockIdx.x, max_tiles, W, jobs.njobs, w, tb, job);
  if (job >= jobs.njobs) return;
  const u32 ntiles = jobs.ntiles[job];
  if (tb >= ntiles) return;
  const u64 n = jobs.n[job];
  const u32* dig = dig_all + jobs.ent_off[job];
  u32* bh = bh_all + jobs.bh_off[job];
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) h[b] = 0;
  __syncthreads();
  const u64 lo = (u64)tb * tile;
  u64 hi = lo + tile; if (hi > n) hi = n;
  const u32* d = dig + (u64)w * n;
  for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    u32 e = d[i] & 0x7fffffffu;
    if (e) atomicAdd(&h[e - 1], 1u);
  }
  __syncthreads();
  u32* out = bh + ((u64)w * ntiles + tb) * nb;
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) out[b] = h[b];
}

// ---- 3. colscan: thread per (w, bucket): exclusive prefix over tiles ----------
__global__ __launch_bounds__(256) void colscan_kernel(Jobs jobs, u32* __restrict__ bh_all, u32* __restrict__ tot_all, u32 nb, u32 W) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  u32 w = blockIdx.y, job = blockIdx.z;
  if (b >= nb) return;
  const u32 ntiles = jobs.ntiles[job];
  u32* bh = bh_all + jobs.bh_off[job];
  u32* tot = tot_all + (u64)job * W * nb;
  u32 run = 0;
  u32* p = bh + (u64)w * ntiles * nb + b;
  for (u32 t = 0; t < ntiles; t++) {
    u32 v = p[(u64)t * nb];
    p[(u64)t * nb] = run;
    run += v;
  }
  tot[(u64)w * nb + b] = run;
}

// ---- 4. binscan: block per window: exclusive scan of tot -> base -----------------
// also folds the largest bucket size of the whole launch into *maxout (selects the accumulation algorithm)
__global__ __launch_bounds__(1024) void binscan_kernel(const u32* __restrict__ tot, u32* __restrict__ base, u32 nb, u32* __restrict__ maxout,
                                                       u32* __restrict__ wtot = nullptr) {
  __shared__ u32 part[1024];
  __shared__ u32 wmax;
  const u32 w = blockIdx.x;
  const u32 per = (nb + 1023) / 1024;
  const u32 lo = threadIdx.x * per;
  u32 s = 0, m = 0;
  if (threadIdx.x == 0) wmax = 0;
  for (u32 k = 0; k < per; k++) { u32 b = lo + k; if (b < nb) { u32 t = tot[(u64)w * nb + b]; s += t; m = t > m ? t : m; } }
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) { u32 o = __shfl_down(m, off); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(&wmax, m);
  part[threadIdx.x] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (u32 off = 1; off < 1024; off <<= 1) {
    u32 v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - s;
  for (u32 k = 0; k < per; k++) {
    u32 b = lo + k;
    if (b < nb) { base[(u64)w * nb + b] = run; run += tot[(u64)w * nb + b]; }
  }
  if (threadIdx.x == 0 && wmax) atomicMax(maxout, wmax);
  if (wtot && threadIdx.x == 1023) wtot[w] = part[1023];      // the window's entries (fixed-base path: woff_kernel turns them into list offsets)
}

// ---- 5. scatter -------------------------------------------------------------------
__global__ __launch_bounds__(HIST_THREADS) void scatter_kernel(Jobs jobs, const u32* __restrict__ dig_all, const u32* __restrict__ bh_all,
                                                               const u32* __restrict__ base_all, u32* __restrict__ sorted_all,
                                                               u32 nb, u32 tile, u32 W, u32 max_tiles) {
  extern __shared__ __attribute__((aligned(16))) u32 h[];
  u32 w, tb, job;
  xcd_decode(blockIdx.x, max_tiles, W, jobs.njobs, w, tb, job);
  if (job >= jobs.njobs) return;
  const u32 ntiles = jobs.ntiles[job];
  if (tb >= ntiles) return;
  const u64 n = jobs.n[job];
  const u32* dig = dig_all + jobs.ent_off[job];
  const u32* bh = bh_all + jobs.bh_off[job];
  const u32* base = base_all + (u64)job * W * nb;
  u32* sorted = sorted_all + jobs.ent_off[job];
  const u32* pre = bh + ((u64)w * ntiles + tb) * nb;
  const u32* bs = base + (u64)w * nb;
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) h[b] = bs[b] + pre[b];
  __syncthreads();
  const u64 lo = (u64)tb * tile;
  u64 hi = lo + tile; if (hi > n) hi = n;
  const u32* d = dig + (u64)w * n;
  u32* out = sorted + (u64)w * n;
  for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    u32 e = d[i];
    u32 b = e & 0x7fffffffu;
    if (b) {
      u32 pos = atomicAdd(&h[b - 1], 1u);
      out[pos] = (u32)i | (e & 0x80000000u);
    }
  }
}

// ---- 6. accum: thread per (window, bucket) ------------------------------------------
// Hot loop: XYZZ accumulator += affine base (8M + 2S), started from the bucket's first entry so the
// accumulator is never the identity.  The two cases the fast formulas cannot handle -- the base equals
// the accumulator (needs a doubling) or its negative (result is the identity) -- are detected by
// P == 0, occur with probability ~2^-381 on honest inputs, and are DEFERRED: the entry is moved to the
// front of the bucket's own list and handled by fixup_kernel with the complete group law.  Keeping the
// slow path (and any call) out of this loop takes the kernel from 248 VGPRs + 304 B of scratch per lane
// to 166 VGPRs and no scratch (3 waves per SIMD).
//
// Scheduling: Fq multiplication throughput saturates at 2 waves per SIMD (profiles/r01_occupancy.txt), so what
// is left to win is idle lanes: bucket sizes are Poisson (sigma ~ 11 at load 128) and a wave runs as long as its
// largest bucket.  Each block therefore sorts the sizes of its ACC_TPB consecutive buckets in LDS (bitonic, key =
// size << 8 | index) and thread t takes the t-th largest, so the 64 lanes of a wave get near-equal trip counts.
// (A persistent variant that handed buckets to lanes dynamically was slower: lanes refilling a bucket stall the
// lanes that are adding, through the dependent loads of the refill path.)
constexpr int ACC_TPB = 256;
// thread -> bucket of the block's ACC_TPB consecutive buckets [lo, lo + ACC_TPB), largest bucket first (see above);
// keys: ACC_TPB words of LDS
__device__ __forceinline__ u64 balanced_bucket(u32* keys, const u32* __restrict__ tot, u64 lo, u64 WB) {
  {
    u64 g = lo + threadIdx.x;
    u32 c = g < WB ? tot[g] : 0;
    if (c > 0xffffffu) c = 0xffffffu;
    keys[threadIdx.x] = (c << 8) | threadIdx.x;
  }
  __syncthreads();
  for (u32 sz = 2; sz <= ACC_TPB; sz <<= 1) {
    for (u32 st = sz >> 1; st > 0; st >>= 1) {
      if (threadIdx.x < ACC_TPB / 2) {
        u32 i = threadIdx.x;
        u32 a = 2 * i - (i & (st - 1)), b = a + st;
        bool desc = ((a & sz) == 0);
        u32 ka = keys[a], kb = keys[b];
        if ((ka < kb) == desc) { keys[a] = kb; keys[b] = ka; }
      }
      __syncthreads();
    }
  }
  return lo + (keys[threadIdx.x] & 255u);
}

__global__ __launch_bounds__(ACC_TPB) void accum_kernel(Jobs jobs, u32* __restrict__ sorted_all,
                                                        const u32* __restrict__ base, const u32* __restrict__ tot,
                                                        G1Xyzz* __restrict__ buckets, u32* __restrict__ pend, u32 nb, u32 W,
                                                        u64 WB) {
  __shared__ u32 keys[ACC_TPB];
  const u64 gid = balanced_bucket(keys, tot, (u64)blockIdx.x * ACC_TPB, WB);
  if (gid >= WB) return;
  const u32 job = (u32)(gid / ((u64)W * nb));
  const u32 w = (u32)((gid / nb) % W);
  const G1Affine* __restrict__ bases = jobs.bases[job];
  u32* lst = sorted_all + jobs.ent_off[job] + (u64)w * jobs.n[job] + base[gid];
  const u32 cnt = tot[gid];
  if (cnt == 0) { g1_store_xyzz(buckets + gid, G1Xyzz::identity()); pend[gid] = 0; return; }
  G1Xyzz acc;
  {
    u32 e = lst[0];
    G1Affine p = g1_load_affine(bases + (e & 0x7fffffffu));
    if (e & 0x80000000u) p.y = ff_neg(p.y);
    acc.x = p.x; acc.y = p.y; acc.zz = Fq::one(); acc.zzz = Fq::one();
  }
  u32 np = 0;
  for (u32 k = 1; k < cnt; k++) {
    u32 e = lst[k];
    G1Affine p = g1_load_affine(bases + (e & 0x7fffffffu));
    if (e & 0x80000000u) p.y = ff_neg(p.y);
    Fq P = ff_sub(ff_mul(p.x, acc.zz), acc.x);
    if (__builtin_expect(P.is_zero(), 0)) { lst[np++] = e; continue; }   // np <= k: never overtakes the read cursor
    Fq R = ff_sub(ff_mul(p.y, acc.zzz), acc.y);
    Fq PP = ff_sqr(P);
    acc.zz = ff_mul(acc.zz, PP);
    Fq Q = ff_mul(acc.x, PP);
    PP = ff_mul(P, PP);            // PPP
    acc.zzz = ff_mul(acc.zzz, PP);
    acc.y = ff_mul(acc.y, PP);     // Y1 * PPP
    Fq X3 = ff_sub(ff_sub(ff_sqr(R), PP), ff_dbl(Q));
    acc.y = ff_sub(ff_mul(R, ff_sub(Q, X3)), acc.y);
    acc.x = X3;
  }
  g1_store_xyzz(buckets + gid, acc);
  pend[gid] = np;
}

// deferred entries (see accum_kernel): full group law, one thread per bucket that has any
__global__ __launch_bounds__(64) void fixup_kernel(Jobs jobs, const u32* __restrict__ sorted_all,
                                                   const u32* __restrict__ base, const u32* __restrict__ pend,
                                                   G1Xyzz* __restrict__ buckets, u32 nb, u32 W, u64 WB) {
  u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= WB) return;
  const u32 np = pend[gid];
  if (np == 0) return;
  const u32 job = (u32)(gid / ((u64)W * nb));
  const u32 w = (u32)((gid / nb) % W);
  const G1Affine* __restrict__ bases = jobs.bases[job];
  const u32* lst = sorted_all + jobs.ent_off[job] + (u64)w * jobs.n[job] + base[gid];
  G1Xyzz acc = g1_load_xyzz(buckets + gid);
  for (u32 k = 0; k < np; k++) {
    u32 e = lst[k];
    G1Affine p = g1_load_affine(bases + (e & 0x7fffffffu));
    if (e & 0x80000000u) p.y = ff_neg(p.y);
    g1_madd(acc, p.x, p.y);
  }
  g1_store_xyzz(buckets + gid, acc);
}

// ---- 7. reduce1: thread per (window, segment) ----------------------------------------
// seg = buckets per segment: SEG on the variable-base path; the fixed-base path shortens the segments when a launch
// would otherwise have fewer threads than the chip has lanes (2 + ~27 / seg additions per bucket, but no idle SIMDs)
__global__ __launch_bounds__(64) void reduce1_kernel(const G1Xyzz* __restrict__ buckets, G1Xyzz* __restrict__ segsum,
                                                     u32 nb, u32 nseg, u32 W, u32 seg) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= W * nseg) return;
  u32 w = gid / nseg, s = gid % nseg;
  u32 lo = s * seg;
  u32 hi = lo + seg; if (hi > nb) hi = nb;
  G1Xyzz running = G1Xyzz::identity(), acc = G1Xyzz::identity();
  const G1Xyzz* B = buckets + (u64)w * nb;
  for (u32 b = hi; b-- > lo;) {
    G1Xyzz t = g1_load_xyzz(B + b);
    g1_add(running, t);
    g1_add(acc, running);
  }
  // + lo * running   (lo = number of buckets below this segment)
  if (lo) {
    G1Xyzz m = G1Xyzz::identity();
    int top = 31 - __clz(lo);
    for (int bit = top; bit >= 0; bit--) {
      g1_dbl(m);
      if ((lo >> bit) & 1) g1_add(m, running);
    }
    g1_add(acc, m);
  }
  g1_store_xyzz(segsum + gid, acc);
}

// ---- 8. reduce2: tree sums ---------------------------------------------------------------
// grid (chunks, windows): block (k, w) sums elements [k * per, (k + 1) * per) of window w's nseg points into
// out[w * chunks + k].  One launch with chunks = 1 when nseg is small, two launches (chunks = nseg / 256, then 1)
// when a single block per window would spend its time in a long serial loop (the fixed-base path: 16384 segments).
__global__ __launch_bounds__(256) void reduce2_kernel(const G1Xyzz* __restrict__ segsum, G1Xyzz* __restrict__ winsum,
                                                      u32 nseg) {
  __shared__ G1Xyzz sh[256];
  const u32 w = blockIdx.y, chunks = gridDim.x;
  const u32 per = (nseg + chunks - 1) / chunks;
  const u32 lo = blockIdx.x * per;
  u32 hi = lo + per; if (hi > nseg) hi = nseg;
  G1Xyzz acc = G1Xyzz::identity();
  for (u32 s = lo + threadIdx.x; s < hi; s += 256) {
    G1Xyzz t = g1_load_xyzz(segsum + (u64)w * nseg + s);
    g1_add(acc, t);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (u32 off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      G1Xyzz a = sh[threadIdx.x];
      G1Xyzz b = sh[threadIdx.x + off];
      g1_add(a, b);
      sh[threadIdx.x] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) g1_store_xyzz(winsum + (u64)w * chunks + blockIdx.x, sh[0]);
}

}  // namespace msm
