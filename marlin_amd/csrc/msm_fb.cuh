// Fixed-base MSM: Pippenger over a precomputed window table, for base sets that are used again and again.
//
// The bases of every KZG10 commitment are the SAME powers of tau (ark-poly-commit kzg10 `commit` on
// `powers_of_g[offset..]`; reference call sites /root/reference src/lib.rs:125,172,193,213,292), so the shifts
// 2^{start_j} * P_i of all W window positions are computed once per base set (`mh_bases_precompute`, 13 x 96 B per
// point at c = 20 -- HBM capacity is what MI355X has plenty of) and every (scalar, window) digit becomes an entry
// "add table point T[j][i] to bucket |digit|".  All windows then share ONE set of 2^(c-1) buckets, the bucket
// reduction no longer scales with the number of windows, and c can grow past the 16 bits an LDS histogram holds:
// c = 20 needs 13 bucket additions per scalar instead of 16.
//
// Sorting 13 n entries by a 19-bit bucket id is done in two levels:
//   count/pscan/pstart/split : partition the entries by the top bits of the bucket id into <= 256 "virtual windows"
//                       of 2^11 buckets (per-block LDS counters, a scan over blocks, then the split proper)
//   hist/colscan/binscan/scatter : a counting sort inside every virtual window, driven by a descriptor (offset,
//                       count, tiles) per virtual window because their sizes differ
// Both split and scatter first sort their block's entries in LDS and then copy the sorted block out, so the lanes
// of a wave store to consecutive addresses: runs of ~200 B per (block, partition) in split, ~32 B per (tile, bucket)
// in scatter.  Ranking straight into global memory (what the variable-base path's scatter does) issues one 4-byte
// store per entry to ~random lines; the L2s cannot keep that many partial lines open, and HBM sees ~8x the payload
// (profiles/pmc_traffic.json; measured the same way for this path before the LDS staging: 2.6 GB written per launch
// for 0.33 GB of entries).
// followed by msm.cuh's accumulate (entries index the table), fix-up and segment reduction with W = 1.
#pragma once
#include "msm.cuh"

namespace msmfb {
using msm::FbWin;
using msm::Windows;

constexpr int MAX_C = 20;          // 2^19 buckets = 256 partitions x 2^11
constexpr int PART_BITS = 11;      // buckets per virtual window = 2^11
constexpr int MAX_PARTS = 256;
inline u32 part_bits(u32 c) { return c - 1 < PART_BITS ? c - 1 : PART_BITS; }
constexpr int TPB = 256;           // count kernel
constexpr int SORT_THREADS = 1024; // split / hist / scatter
constexpr int SPLIT_ENTRIES = 13312;   // entries one split block stages in LDS (13 windows x 1024 scalars)
constexpr int TILE_EPT = 16;           // scatter: entries per thread -> tiles of <= 16384 entries
constexpr int MAX_TILE = TILE_EPT * SORT_THREADS;
// scalars per count/split block for W windows
inline u32 split_scalars(u32 W) { u32 sc = (SPLIT_ENTRIES / W) & ~63u; return sc > 1024 ? 1024 : sc; }
inline size_t split_lds_bytes(u32 W) { return (size_t)(2 * MAX_PARTS + 32) * 4 + (size_t)split_scalars(W) * W * 9 + 16; }
inline size_t scatter_lds_bytes(u32 nb) { return (size_t)(2 * nb + 32) * 4 + (size_t)MAX_TILE * 6; }

// exclusive scan of a[0, n) in LDS for n <= 2 * blockDim.x (blockDim.x a multiple of 64); tmp: 32 words of LDS
__device__ __forceinline__ void block_excl_scan2(u32* a, u32 n, u32* tmp) {
  const u32 t = threadIdx.x;
  const u32 x0 = 2 * t < n ? a[2 * t] : 0, x1 = 2 * t + 1 < n ? a[2 * t + 1] : 0;
  const u32 sum = x0 + x1;
  u32 inc = sum;
  for (int off = 1; off < 64; off <<= 1) { u32 o = __shfl_up(inc, off); if ((t & 63) >= (u32)off) inc += o; }
  if ((t & 63) == 63) tmp[t >> 6] = inc;
  __syncthreads();
  if (t < 64) {
    const u32 nw = blockDim.x >> 6;
    const u32 v = t < nw ? tmp[t] : 0;
    u32 iv = v;
    for (int off = 1; off < 64; off <<= 1) { u32 o = __shfl_up(iv, off); if (t >= (u32)off) iv += o; }
    if (t < nw) tmp[t] = iv - v;
  }
  __syncthreads();
  const u32 excl = inc - sum + tmp[t >> 6];
  if (2 * t < n) a[2 * t] = excl;
  if (2 * t + 1 < n) a[2 * t + 1] = excl + x0;
  __syncthreads();
}

struct FbJobs {
  const Fr* scalars[msm::MAX_JOBS];
  u64 n[msm::MAX_JOBS];
  u64 ent_off[msm::MAX_JOBS];    // job's region in key / val / sorted (W * n entries)
  u64 pc_off[msm::MAX_JOBS];     // job's region in the per-block partition counts (nparts * nblk)
  u32 tab_off[msm::MAX_JOBS];    // index of the job's first base inside the table's base set
  u32 nblk[msm::MAX_JOBS];
  u32 njobs;
};

// ---- table: level j from level j-1 by `bits` doublings, back to affine ---------------------------------
__global__ __launch_bounds__(128) void table_level_kernel(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next, u64 n, u32 bits) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = g1_load_affine(prev + i);
  G1Xyzz a;
  g1_dbl_affine(a, p.x, p.y);
  for (u32 k = 1; k < bits; k++) g1_dbl(a);
  // x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2: 1/ZZ = ZZ^2 / ZZZ^2
  Fq iz = ff_inv(a.zzz);
  Fq izz = ff_mul(ff_sqr(a.zz), ff_sqr(iz));
  G1Affine r;
  r.x = ff_mul(a.x, izz);
  r.y = ff_mul(a.y, iz);
  ff_store(&next[i].x, r.x);
  ff_store(&next[i].y, r.y);
}

// ---- partition pass ------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void count_kernel(FbJobs jobs, u32* __restrict__ pc, u32 W, Windows win, int is_mont, u32 nparts,
                                                    u32 pshift, u32 S) {
  __shared__ u32 cnt[MAX_PARTS];
  const u32 job = blockIdx.y, blk = blockIdx.x;
  if (blk >= jobs.nblk[job]) return;
  static_assert(TPB >= MAX_PARTS, "one thread per partition counter");
  if (threadIdx.x < MAX_PARTS) cnt[threadIdx.x] = 0;
  __syncthreads();
  const u64 n = jobs.n[job];
  for (u32 k = threadIdx.x; k < S; k += TPB) {
    u64 i = (u64)blk * S + k;
    if (i < n) {
      Fr s = ff_load(jobs.scalars[job] + i);
      if (is_mont) s = ff_from_mont(s);
      msm::for_each_digit(s, W, win, [&](u32, u32 e) {
        u32 b = e & 0x7fffffffu;
        if (b) atomicAdd(&cnt[(b - 1) >> pshift], 1u);
      });
    }
  }
  __syncthreads();
  if (threadIdx.x < nparts) pc[jobs.pc_off[job] + (u64)threadIdx.x * jobs.nblk[job] + blk] = cnt[threadIdx.x];
}

// block per (partition, job): exclusive scan of the per-block counts in place, total -> ptot[job * nparts + v]
__global__ __launch_bounds__(1024) void pscan_kernel(FbJobs jobs, u32* __restrict__ pc, u32* __restrict__ ptot, u32 nparts) {
  __shared__ u32 part[1024];
  const u32 v = blockIdx.x, job = blockIdx.y;
  const u32 nblk = jobs.nblk[job];
  u32* row = pc + jobs.pc_off[job] + (u64)v * nblk;
  const u32 per = (nblk + 1023) / 1024;
  const u32 lo = threadIdx.x * per;
  u32 s = 0;
  for (u32 k = 0; k < per; k++) if (lo + k < nblk) s += row[lo + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    u32 t = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - s;
  for (u32 k = 0; k < per; k++)
    if (lo + k < nblk) { u32 t = row[lo + k]; row[lo + k] = run; run += t; }
  if (threadIdx.x == 1023) ptot[job * nparts + v] = part[1023];
}

// block per job: pstart[job * nparts + v] = sum of the job's partition totals below v
__global__ __launch_bounds__(MAX_PARTS) void pstart_kernel(const u32* __restrict__ ptot, u32* __restrict__ pstart, u32 nparts) {
  __shared__ u32 sh[MAX_PARTS];
  const u32 job = blockIdx.x, v = threadIdx.x;
  const u32 mine = v < nparts ? ptot[job * nparts + v] : 0;
  sh[v] = mine;
  __syncthreads();
  for (u32 off = 1; off < MAX_PARTS; off <<= 1) {
    u32 t = v >= off ? sh[v - off] : 0;
    __syncthreads();
    sh[v] += t;
    __syncthreads();
  }
  if (v < nparts) pstart[job * nparts + v] = sh[v] - mine;
}

// One block = S scalars (the same S as count_kernel): digits -> LDS counters -> local offsets -> entries placed in
// LDS grouped by partition -> copied out, consecutive lanes to consecutive addresses of each partition's run.
__global__ __launch_bounds__(SORT_THREADS) void split_kernel(FbJobs jobs, const u32* __restrict__ pc, const u32* __restrict__ pstart,
                                                             u32* __restrict__ key, u32* __restrict__ val, u32 W, Windows win,
                                                             int is_mont, u32 nparts, u32 pshift, u32 tab_n, u32 S) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* cur = lds;                                  // [MAX_PARTS] counts -> local starts -> cursors
  u32* gdst = lds + MAX_PARTS;                     // [MAX_PARTS] global position of local entry 0 of each partition
  u32* tmp = lds + 2 * MAX_PARTS;                  // [32]
  u32* skey = lds + 2 * MAX_PARTS + 32;            // [S * W]
  u32* sval = skey + S * W;
  unsigned char* spart = (unsigned char*)(sval + S * W);
  const u32 job = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  if (blk >= jobs.nblk[job]) return;
  if (t < MAX_PARTS) cur[t] = 0;
  __syncthreads();
  const u64 n = jobs.n[job];
  const u64 i = (u64)blk * S + t;
  const bool live = t < S && i < n;
  Fr s;
  if (live) {
    s = ff_load(jobs.scalars[job] + i);
    if (is_mont) s = ff_from_mont(s);
    msm::for_each_digit(s, W, win, [&](u32, u32 e) {
      u32 b = e & 0x7fffffffu;
      if (b) atomicAdd(&cur[(b - 1) >> pshift], 1u);
    });
  }
  __syncthreads();
  block_excl_scan2(cur, MAX_PARTS, tmp);
  if (t < nparts) gdst[t] = pstart[job * nparts + t] + pc[jobs.pc_off[job] + (u64)t * jobs.nblk[job] + blk] - cur[t];
  __syncthreads();
  if (live) {
    const u32 t0 = jobs.tab_off[job] + (u32)i;
    msm::for_each_digit(s, W, win, [&](u32 w, u32 e) {
      u32 b = e & 0x7fffffffu;
      if (b) {
        b -= 1;
        const u32 v = b >> pshift;
        const u32 p = atomicAdd(&cur[v], 1u);
        skey[p] = ((b & ((1u << pshift) - 1)) + 1) | (e & 0x80000000u);
        sval[p] = w * tab_n + t0;
        spart[p] = (unsigned char)v;
      }
    });
  }
  __syncthreads();
  const u32 total = cur[MAX_PARTS - 1];             // cursor of the last partition = number of entries of the block
  u32* kj = key + jobs.ent_off[job];
  u32* vj = val + jobs.ent_off[job];
  for (u32 idx = t; idx < total; idx += SORT_THREADS) {
    const u32 d = gdst[spart[idx]] + idx;
    kj[d] = skey[idx];
    vj[d] = sval[idx];
  }
}

// ---- counting sort inside the virtual windows ------------------------------------------------------------
// Block -> (virtual window, tile) comes from a host-built list, XCD-interleaved: block b runs on XCD b % 8 (see
// msm::xcd_decode) and XCD x works through the virtual windows x, x + 8, x + 16, ... one after another, so that the
// partial lines of one window's bucket lists meet in that XCD's L2.  gw = ~0 marks padding.
struct FbBlk { u32 gw, tile; };

__global__ __launch_bounds__(SORT_THREADS) void hist_kernel(const FbWin* __restrict__ fbw, const FbBlk* __restrict__ blk,
                                                                 const u32* __restrict__ key, u32* __restrict__ bh, u32 nb, u32 tile) {
  extern __shared__ __attribute__((aligned(16))) u32 h[];
  const u32 gw = blk[blockIdx.x].gw, tb = blk[blockIdx.x].tile;
  if (gw == 0xffffffffu) return;
  const FbWin d = fbw[gw];
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) h[b] = 0;
  __syncthreads();
  const u32 lo = tb * tile;
  u32 hi = lo + tile; if (hi > d.cnt) hi = d.cnt;
  const u32* k = key + d.off;
  for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&h[(k[i] & 0x7fffffffu) - 1], 1u);
  __syncthreads();
  u32* out = bh + d.bh_off + (u64)tb * nb;
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) out[b] = h[b];
}

// thread per (virtual window, bucket): exclusive prefix over the window's tiles + bucket totals
__global__ __launch_bounds__(256) void colscan_kernel(const FbWin* __restrict__ fbw, u32* __restrict__ bh, u32* __restrict__ tot, u32 nb) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 gw = blockIdx.y;
  if (b >= nb) return;
  const FbWin d = fbw[gw];
  u32 run = 0;
  u32* p = bh + d.bh_off + b;
  for (u32 t = 0; t < d.ntiles; t++) {
    u32 v = p[(u64)t * nb];
    p[(u64)t * nb] = run;
    run += v;
  }
  tot[(u64)gw * nb + b] = run;
}

// One block = one tile of <= MAX_TILE entries of one virtual window: local ranks by LDS atomics, local bucket starts by
// a scan, entries placed in LDS in bucket order, then copied out (global position = where the tile's share of the
// bucket starts + offset inside the tile's share).
__global__ __launch_bounds__(SORT_THREADS) void scatter_kernel(const FbWin* __restrict__ fbw, const FbBlk* __restrict__ blk,
                                                               const u32* __restrict__ key, const u32* __restrict__ val,
                                                               const u32* __restrict__ bh, const u32* __restrict__ base,
                                                               u32* __restrict__ sorted, u32 nb, u32 tile) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* lcnt = lds;                                 // [nb] counts -> local starts
  u32* gdst = lds + nb;                            // [nb]
  u32* tmp = lds + 2 * nb;                         // [32]
  u32* stage = lds + 2 * nb + 32;                  // [MAX_TILE]
  unsigned short* sbkt = (unsigned short*)(stage + MAX_TILE);
  const u32 gw = blk[blockIdx.x].gw, tb = blk[blockIdx.x].tile, t = threadIdx.x;
  if (gw == 0xffffffffu) return;
  const FbWin d = fbw[gw];
  for (u32 b = t; b < nb; b += SORT_THREADS) lcnt[b] = 0;
  __syncthreads();
  const u32 lo = tb * tile;
  const u32 cnt = d.cnt - lo < tile ? d.cnt - lo : tile;
  const u32* k = key + d.off + lo;
  const u32* v = val + d.off + lo;
  u32 rk[TILE_EPT], ev[TILE_EPT], bk[TILE_EPT];
#pragma unroll
  for (int j = 0; j < TILE_EPT; j++) {
    const u32 i = j * SORT_THREADS + t;
    if (i < cnt) {
      const u32 e = k[i];
      bk[j] = (e & 0x7fffffffu) - 1;
      ev[j] = v[i] | (e & 0x80000000u);
      rk[j] = atomicAdd(&lcnt[bk[j]], 1u);
    }
  }
  __syncthreads();
  block_excl_scan2(lcnt, nb, tmp);
  {
    const u32* pre = bh + d.bh_off + (u64)tb * nb;
    const u32* bs = base + (u64)gw * nb;
    for (u32 b = t; b < nb; b += SORT_THREADS) gdst[b] = bs[b] + pre[b] - lcnt[b];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TILE_EPT; j++) {
    const u32 i = j * SORT_THREADS + t;
    if (i < cnt) {
      const u32 idx = lcnt[bk[j]] + rk[j];
      stage[idx] = ev[j];
      sbkt[idx] = (unsigned short)bk[j];
    }
  }
  __syncthreads();
  u32* out = sorted + d.off;
  for (u32 idx = t; idx < cnt; idx += SORT_THREADS) out[gdst[sbkt[idx]] + idx] = stage[idx];
}

}  // namespace msmfb
