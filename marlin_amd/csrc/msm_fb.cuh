// Fixed-base MSM: Pippenger over a precomputed window table, for base sets that are used again and again.
//
// The bases of every KZG10 commitment are the SAME powers of tau (ark-poly-commit kzg10 `commit` on
// `powers_of_g[offset..]`; reference call sites /root/reference src/lib.rs:125,172,193,213,292), so the shifts
// 2^{start_j} * P_i of all W window positions are computed once per base set (`mh_bases_precompute`, 13 x 128 B per
// point at c = 20 -- HBM capacity is what MI355X has plenty of) and every (scalar, window) digit becomes an entry
// "add table point T[j][i] to bucket |digit|".  All windows then share ONE set of 2^(c-1) buckets, the bucket
// reduction no longer scales with the number of windows, and c can grow past the 16 bits an LDS histogram holds:
// c = 20 needs 13 bucket additions per scalar instead of 16.  Table points, bucket accumulators and the bucket
// reduction use the 30-bit-limb field arithmetic of fq30.cuh.
//
// Sorting 13 n entries by a 19-bit bucket id is done in two levels:
//   psplit            : ONE pass over the scalars partitions the entries by the top bits of the bucket id into <= 256 "virtual
//                       windows" of 2^11 buckets -- per split block, into the block's own region of the entry arrays (round 6;
//                       rounds 1-5: count / pscan / pstart / split, two passes and a host round trip for the totals)
//   hist/colscan/binscan/woff/scatter : a counting sort inside every virtual window over tiles of ~13 K entries (a tile = the
//                       window's runs in consecutive split blocks), the windows' list offsets computed on the device
// Both psplit and scatter first sort their block's entries in LDS and then copy the sorted block out, so the lanes
// of a wave store to consecutive addresses: the whole block in psplit, ~32 B per (tile, bucket)
// in scatter.  Ranking straight into global memory (what the variable-base path's scatter does) issues one 4-byte
// store per entry to ~random lines; the L2s cannot keep that many partial lines open, and HBM sees ~8x the payload
// (profiles/pmc_traffic.json; measured the same way for this path before the LDS staging: 2.6 GB written per launch
// for 0.33 GB of entries).  No stage of the sort waits for the host: the block list and the descriptors' host part depend on the
// batch's shape only.
// followed by the ordering of all buckets by size, the accumulate kernel (entries index the table), the fix-up of deferred
// collisions, and the reduction of the single bucket set of each MSM (row / column sums, bit planes).
#pragma once
#include "msm.cuh"
#include "x30.cuh"

namespace msmfb {
using msm::Windows;

// An entry between the two sort levels is a 16-bit key (its bucket inside the virtual window) and a 32-bit value (table index,
// sign of the digit in bit 31): 6 bytes written by split, 2 read by hist, 6 by scatter -- the stages run at HBM / LDS-atomic
// speed, so bytes are time (32-bit keys with the sign in the key: 8 / 4 / 8).
// Virtual window: a variable-length run of the entry arrays holding the entries of 2^PART_BITS consecutive buckets
// delta: added to every table index of the window's entries -- 0 except for a job that SHARES the sorted lists of another
// job with the same scalars and a base range `delta` points further into the same base set (MarlinKZG10 commits a
// degree-bounded polynomial twice: against powers and against shifted_powers(d) = powers[max_degree - d ..])
// off / cnt: where the window's bucket lists start in `sorted` and how many entries they hold -- written on the DEVICE (woff_kernel);
// bh_off / ntiles / bpt (split blocks per tile) / delta come from the host and depend on the batch's shape only
struct FbWin { u64 off; u64 bh_off; u32 cnt; u32 ntiles; u32 delta; u32 bpt; };

constexpr int MAX_C = 20;          // 2^19 buckets = 256 partitions x 2^11
constexpr int PART_BITS = 11;      // buckets per virtual window = 2^11
constexpr int MAX_PARTS = 256;
inline u32 part_bits(u32 c) { return c - 1 < PART_BITS ? c - 1 : PART_BITS; }
constexpr int TPB = 1024;          // count kernel: one scalar per thread
constexpr int SORT_THREADS = 1024; // split / hist / scatter
constexpr int SPLIT_ENTRIES = 13312;   // entries one split block stages in LDS (13 windows x 1024 scalars)
constexpr int MAX_TILE = 16 * SORT_THREADS;   // entries one hist / scatter tile stages in LDS
// scalars per count/split block for W windows
inline u32 split_scalars(u32 W) { u32 sc = (SPLIT_ENTRIES / W) & ~63u; return sc > 1024 ? 1024 : sc; }
inline size_t split_lds_bytes(u32 W) { return (size_t)(MAX_PARTS + 32) * 4 + (size_t)split_scalars(W) * W * 6 + 16; }
inline size_t scatter_lds_bytes(u32 nb) { return (size_t)(4 * nb + 32) * 4 + (size_t)MAX_TILE * 6 + 4096 * 4; }
inline size_t hist_lds_bytes(u32 nb) { return (size_t)nb * 4 + 4096 * 4; }

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
  u64 ent_off[msm::MAX_JOBS];    // job's region in key / val (nblk blocks of S * W entries)
  u64 lst_off[msm::MAX_JOBS];    // job's region in the table of per-block partition starts (nblk * (nparts + 1) offsets)
  u32 tab_off[msm::MAX_JOBS];    // index of the job's first base inside the table's base set
  u32 tab_stride[msm::MAX_JOBS]; // scalar i multiplies base tab_off + i * tab_stride (1: a contiguous range; G: rank's cyclic slice)
  u32 nblk[msm::MAX_JOBS];
  u32 njobs;
};

// ---- table -----------------------------------------------------------------------------------------------
// level j from level j-1 by `bits` doublings; the standard-form result feeds the next level, the 30-bit-limb form is what
// the accumulation reads.  Level 0 (bits = 0) is NOT the point itself but [R^-1 mod r] P (R = 2^256, kinv = that
// integer): the prover's scalars arrive in Montgomery form -- the integer s R mod r -- and sum (s_i R) ([R^-1] P_i) =
// sum s_i P_i, so the partition kernels recode the words they load and never run a Montgomery reduction per scalar and
// window pass (canonical scalars are multiplied by R instead, at the cost the reduction had).
// One thread takes TAB_BATCH points (i, i + T, i + 2T, ... with T = ceil(n / TAB_BATCH), so the loads of a wave stay
// coalesced) and shares ONE field inversion among them (Montgomery's trick: prefix products of the ZZZ's in registers,
// the un-normalised X, Y, ZZ, ZZZ parked in `scratch`): per point 20 doublings + ~8 multiplications + 1/8 of a Fermat
// inversion instead of a whole one (~570 multiplications), which was three quarters of the table build.
constexpr int TAB_BATCH = 8;
struct G1XyzzStd { Fq x, y, zz, zzz; };
__global__ __launch_bounds__(128) void table_level_kernel(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next_std,
                                                          G1Aff30* __restrict__ next30, G1XyzzStd* __restrict__ scratch, u64 n,
                                                          u32 bits, Fr kinv) {
  const u64 T = (n + TAB_BATCH - 1) / TAB_BATCH;
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  Fq pre[TAB_BATCH];                                     // pre[j] = zzz_0 ... zzz_j over this thread's points
  int cnt = 0;
  for (int j = 0; j < TAB_BATCH; j++) {
    const u64 i = t + (u64)j * T;
    if (i >= n) break;
    const G1Affine p = g1_load_affine(prev + i);
    G1Xyzz a;
    if (bits == 0) {
      a = G1Xyzz::identity();
      for (int limb = Fr::N - 1; limb >= 0; limb--)
        for (int b = 31; b >= 0; b--) {
          g1_dbl(a);
          if ((kinv.v[limb] >> b) & 1u) g1_madd(a, p.x, p.y);
        }
    } else {
      g1_dbl_affine(a, p.x, p.y);
      for (u32 k = 1; k < bits; k++) g1_dbl(a);
    }
    // [k] P of a point of prime order r with 0 < k < r is never the identity: ZZZ != 0
    ff_store(&scratch[i].x, a.x); ff_store(&scratch[i].y, a.y); ff_store(&scratch[i].zz, a.zz); ff_store(&scratch[i].zzz, a.zzz);
    pre[j] = j ? ff_mul(pre[j - 1], a.zzz) : a.zzz;
    cnt = j + 1;
  }
  Fq inv = ff_inv(pre[cnt - 1]);
  for (int j = cnt - 1; j >= 0; j--) {
    const u64 i = t + (u64)j * T;
    const Fq zzz = ff_load(&scratch[i].zzz);
    const Fq iz = j ? ff_mul(inv, pre[j - 1]) : inv;     // 1 / zzz_j
    if (j) inv = ff_mul(inv, zzz);
    // x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2: 1 / ZZ = ZZ^2 / ZZZ^2
    const Fq izz = ff_mul(ff_sqr(ff_load(&scratch[i].zz)), ff_sqr(iz));
    const Fq x = ff_mul(ff_load(&scratch[i].x), izz), y = ff_mul(ff_load(&scratch[i].y), iz);
    ff_store(&next_std[i].x, x);
    ff_store(&next_std[i].y, y);
    store30(next30[i].x, f30_from_fq(x));
    store30(next30[i].y, f30_from_fq(y));
  }
}

// ---- partition pass ------------------------------------------------------------------------------------
// Own: the partitions (virtual windows of 2^PART_BITS buckets) this rank owns -- all of them on one GPU ({0, 1}); with the
// MSM sharded by bucket range over G ranks rank g owns the partitions v = g (mod G): every rank recodes
// every scalar but keeps only the digits that fall into its partitions.  Interleaved, not contiguous, because the
// buckets are not equally loaded: the windows narrower than c bits (and the top window) only reach the low buckets.
struct Own { u32 first, stride; };
// (rank counts are powers of two except in tests: the mask form spares the recoding passes a 32-bit division per digit)
__device__ __forceinline__ bool owns(const Own& o, u32 v) {
  return (o.stride & (o.stride - 1)) == 0 ? (v & (o.stride - 1)) == o.first : v % o.stride == o.first;
}
// One pass over the scalars (round 6; rounds 1-5 counted first -- count / pscan / pstart -- and split second, reading and recoding
// every scalar twice, with a host round trip for the partition totals in between).  A block takes S scalars, recodes them ONCE from
// registers in two phases -- LDS counters per partition, a scan, then the entries placed in LDS grouped by partition -- and writes
// its <= S W entries out to ITS OWN fixed region of the entry arrays (block b of a job: entries [b S W, (b + 1) S W)), fully
// coalesced, followed by the table of its partition starts (lst: nparts + 1 offsets).  No global position depends on another
// block, so nothing is counted beforehand and the host learns nothing: a virtual window is no longer one contiguous run but one
// run per block (~50 entries at c = 20), which the tiles of hist / scatter walk block by block.
// C: the table's window width when it is the usual one (20: the recoding is unrolled over compile-time windows, see
// msm::for_each_digit_c), 0: any width (the windows come from `win`)
template <int C>
__global__ __launch_bounds__(SORT_THREADS) void psplit_kernel(FbJobs jobs, unsigned short* __restrict__ key, u32* __restrict__ val,
                                                              unsigned short* __restrict__ lst_all, u32 W, Windows win, int is_mont,
                                                              u32 nparts, u32 pshift, u32 tab_n, u32 S, Own own) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* cur = lds;                                  // [MAX_PARTS] counts -> local starts -> cursors
  u32* tmp = lds + MAX_PARTS;                      // [32]
  u32* sval = lds + MAX_PARTS + 32;                // [S * W] table index | sign << 31
  unsigned short* skey = (unsigned short*)(sval + S * W);   // [S * W] bucket inside the virtual window (< 2^PART_BITS)
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
    if (!is_mont) s = ff_to_mont(s);              // the table holds [R^-1] P: the digits of s R are what multiplies it
    auto count = [&](u32, u32 e) {
      const u32 b = e & 0x7fffffffu;
      if (b) {
        const u32 v = (b - 1) >> pshift;
        if (owns(own, v)) atomicAdd(&cur[v], 1u);
      }
    };
    if constexpr (C > 0) msm::for_each_digit_c<C>(s, count); else msm::for_each_digit(s, W, win, count);
  }
  __syncthreads();
  block_excl_scan2(cur, MAX_PARTS, tmp);
  if (live) {
    const u32 t0 = jobs.tab_off[job] + (u32)i * jobs.tab_stride[job];
    auto place = [&](u32 w, u32 e) {
      u32 b = e & 0x7fffffffu;
      if (b) {
        b -= 1;
        const u32 v = b >> pshift;
        if (!owns(own, v)) return;                  // another rank's partition
        const u32 p = atomicAdd(&cur[v], 1u);
        skey[p] = (unsigned short)(b & ((1u << pshift) - 1));
        sval[p] = (w * tab_n + t0) | (e & 0x80000000u);          // table indices stay below 2^31 (mh_bases_precompute checks)
      }
    };
    if constexpr (C > 0) msm::for_each_digit_c<C>(s, place); else msm::for_each_digit(s, W, win, place);
  }
  __syncthreads();
  // cur[v] is now the END of partition v's run = the start of partition v + 1's
  unsigned short* lst = lst_all + jobs.lst_off[job] + (u64)blk * (nparts + 1);
  if (t == 0) lst[0] = 0;
  if (t < nparts) lst[t + 1] = (unsigned short)cur[t];
  const u32 total = cur[nparts - 1];
  const u64 base = jobs.ent_off[job] + (u64)blk * S * W;
  for (u32 idx = t; idx < total; idx += SORT_THREADS) {
    key[base + idx] = skey[idx];
    val[base + idx] = sval[idx];
  }
}

// ---- counting sort inside the virtual windows ------------------------------------------------------------
// A tile = the runs of ONE virtual window in `bpt` consecutive split blocks (bpt chosen per window from the expected load so that a
// tile holds ~13 K entries; the scatter works an overfull tile off in sub-tiles).  Block -> (virtual window, tile) comes from a host-built list, XCD-interleaved: block b
// runs on XCD b % 8 (see msm::xcd_decode) and XCD x works through the virtual windows x, x + 8, x + 16, ... one after another, so
// that the partial lines of one window's bucket lists meet in that XCD's L2.  gw = ~0 marks padding.  The list depends on the
// batch's shape only -- not on its scalars -- so nothing of it waits for the device.
struct FbBlk { u32 gw, tile; };

// the runs of window (job, v) in split blocks [sb, se) of a tile that starts at block b0.  `lpr` adjacent lanes share a run and a lane
// takes its entries RUN_BATCH at a time: load(j, entry index) for the whole batch first -- independent loads, all in flight --,
// then use(j) for each of them.
constexpr int RUN_BATCH = 4;
constexpr int MAX_BPT = 4096;                        // split blocks per tile (their run starts and lengths are staged in LDS as 16-bit numbers)
// stage the tile's runs (start inside the block, length) in LDS; returns this thread's share of the entries
__device__ __forceinline__ u32 stage_runs(unsigned short* rlo, unsigned short* rlen, const unsigned short* __restrict__ lst, u32 nparts, u32 v, u32 b0, u32 b1) {
  u32 mine = 0;
  for (u32 b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
    const unsigned short* L = lst + (u64)b * (nparts + 1);
    const u32 lo = L[v], r = (u32)L[v + 1] - lo;
    rlo[b - b0] = (unsigned short)lo; rlen[b - b0] = (unsigned short)r; mine += r;
  }
  return mine;
}
template <class LD, class US>
__device__ __forceinline__ void for_each_run_entry(const unsigned short* rlo, const unsigned short* rlen, u64 ent_base, u32 SW, u32 b0, u32 sb, u32 se, LD load, US use) {
  const u32 span = se - sb;
  // lanes per run: at least 16 -- a run is ~100 B of keys and ~200 B of values, and 16 lanes take it in 3-4 coalesced steps (4 lanes
  // per run, all runs in flight at once, measured 0.9 ms per proof slower at 2^20: every load instruction then touches 16 lines;
  // profiles/r06h_sweep_lanes_per_run.txt) --, more when the tile has few runs
  u32 lg = 4;
  while (lg < 6 && ((2u << lg) * span) <= blockDim.x) lg++;
  const u32 lpr = 1u << lg, grp = threadIdx.x >> lg, sub = threadIdx.x & (lpr - 1), ngrp = blockDim.x >> lg;
  for (u32 b = sb + grp; b < se; b += ngrp) {
    const u32 lo = rlo[b - b0], hi = lo + rlen[b - b0];
    const u64 at = ent_base + (u64)b * SW;
    for (u32 i0 = lo + sub; i0 < hi; i0 += lpr * RUN_BATCH) {
#pragma unroll
      for (int j = 0; j < RUN_BATCH; j++) { const u32 i = i0 + j * lpr; if (i < hi) load(j, at + i); }
#pragma unroll
      for (int j = 0; j < RUN_BATCH; j++) { const u32 i = i0 + j * lpr; if (i < hi) use(j); }
    }
  }
}

__global__ __launch_bounds__(SORT_THREADS) void hist_kernel(const FbWin* __restrict__ fbw, const FbBlk* __restrict__ blk, FbJobs jobs,
                                                            const unsigned short* __restrict__ key, const unsigned short* __restrict__ lst_all,
                                                            u32* __restrict__ bh, u32 nb, u32 nparts, u32 SW) {
  extern __shared__ __attribute__((aligned(16))) u32 h[];        // [nb] counters, then the staged runs
  unsigned short* rlo = (unsigned short*)(h + nb);
  unsigned short* rlen = rlo + MAX_BPT;
  const u32 gw = blk[blockIdx.x].gw, tb = blk[blockIdx.x].tile;
  if (gw == 0xffffffffu) return;
  const FbWin d = fbw[gw];
  const u32 job = gw / nparts, v = gw % nparts;
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) h[b] = 0;
  const u32 b0 = tb * d.bpt, b1 = b0 + d.bpt < jobs.nblk[job] ? b0 + d.bpt : jobs.nblk[job];
  (void)stage_runs(rlo, rlen, lst_all + jobs.lst_off[job], nparts, v, b0, b1);
  __syncthreads();
  u32 kk[RUN_BATCH];
  for_each_run_entry(rlo, rlen, jobs.ent_off[job], SW, b0, b0, b1, [&](int j, u64 at) { kk[j] = key[at]; }, [&](int j) { atomicAdd(&h[kk[j]], 1u); });
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

// one block: where each virtual window's bucket lists start in `sorted` (exclusive scan of the windows' entry counts, which
// binscan left in wtot) and how many entries it has -- what the descriptors of rounds 1-5 got from the host after a round trip.
// A window that shares another job's lists (src[gw] != ~0) takes that window's numbers.
__global__ __launch_bounds__(1024) void woff_kernel(FbWin* __restrict__ fbw, const u32* __restrict__ wtot, const u32* __restrict__ src, u32 WT) {
  __shared__ u32 part[1024];
  const u32 per = (WT + 1023) / 1024, lo = threadIdx.x * per;
  u32 s = 0;
  for (u32 k = 0; k < per; k++) if (lo + k < WT && src[lo + k] == 0xffffffffu) s += wtot[lo + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    const u32 t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - s;
  for (u32 k = 0; k < per; k++) {
    const u32 gw = lo + k;
    if (gw < WT && src[gw] == 0xffffffffu) { fbw[gw].off = run; fbw[gw].cnt = wtot[gw]; run += wtot[gw]; }
  }
  __threadfence_block();
  __syncthreads();
  for (u32 k = 0; k < per; k++) {
    const u32 gw = lo + k;
    if (gw < WT && src[gw] != 0xffffffffu) { fbw[gw].off = fbw[src[gw]].off; fbw[gw].cnt = fbw[src[gw]].cnt; }
  }
}

// One block = one tile: bucket counts by LDS atomics over the tile's runs, local bucket starts by a scan, the entries placed in
// LDS in bucket order (second walk over the runs: keys and values come out of L2), then copied out (global position = where the
// tile's share of the bucket starts + offset inside the tile's share), consecutive lanes to consecutive addresses.  The tiles are
// sized on the host for uniformly distributed scalars (~13 K entries); a tile that holds more than the staging area takes (digits
// that repeat: a window whose top digits are all small, a polynomial with a dominant coefficient) is worked off in SUB-TILES of
// consecutive split blocks, `done` carrying every bucket's fill level from one to the next -- no input makes the sort leave this
// path, only the buckets' own sizes do (skew_limit).
static_assert(SPLIT_ENTRIES <= MAX_TILE, "one split block's run must fit a scatter sub-tile");
__global__ __launch_bounds__(SORT_THREADS) void scatter_kernel(const FbWin* __restrict__ fbw, const FbBlk* __restrict__ blk, FbJobs jobs,
                                                               const unsigned short* __restrict__ key, const u32* __restrict__ val,
                                                               const unsigned short* __restrict__ lst_all, const u32* __restrict__ bh,
                                                               const u32* __restrict__ base, const u32* __restrict__ tot, u32* __restrict__ sorted,
                                                               u32 nb, u32 nparts, u32 SW) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* lcnt = lds;                                 // [nb] counts -> local starts
  u32* cur = lds + nb;                             // [nb] cursors
  u32* gdst = lds + 2 * nb;                        // [nb]
  u32* done = lds + 3 * nb;                        // [nb] entries of each bucket the earlier sub-tiles of this tile have written
  u32* tmp = lds + 4 * nb;                         // [32]; tmp[30], tmp[31]: end block and entries of the sub-tile
  u32* stage = lds + 4 * nb + 32;                  // [MAX_TILE]
  unsigned short* sbkt = (unsigned short*)(stage + MAX_TILE);      // [MAX_TILE]
  unsigned short* rlen = sbkt + MAX_TILE;                           // [MAX_BPT] run lengths of the tile's blocks
  unsigned short* rlo = rlen + MAX_BPT;                             // [MAX_BPT] where each run starts inside its block
  const u32 gw = blk[blockIdx.x].gw, tb = blk[blockIdx.x].tile, t = threadIdx.x;
  if (gw == 0xffffffffu) return;
  const FbWin d = fbw[gw];
  const u32 job = gw / nparts, v = gw % nparts;
  const u32 b0 = tb * d.bpt, b1 = b0 + d.bpt < jobs.nblk[job] ? b0 + d.bpt : jobs.nblk[job];
  const unsigned short* lst = lst_all + jobs.lst_off[job];
  for (u32 b = t; b < nb; b += SORT_THREADS) done[b] = 0;
  if (t == 0) tmp[29] = 0;                           // the tile's entries
  __syncthreads();
  {
    u32 mine = stage_runs(rlo, rlen, lst, nparts, v, b0, b1);
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((t & 63) == 0 && mine) atomicAdd(&tmp[29], mine);
  }
  const u32* pre = bh + d.bh_off + (u64)tb * nb;
  const u32* bs = base + (u64)gw * nb;
  u32* out = sorted + d.off;
  const u32* tot_w = tot + (u64)gw * nb;
  const bool last_tile = tb + 1 == d.ntiles;
  u32 kk[RUN_BATCH], vv[RUN_BATCH];
  u32 sb = b0;
  bool whole = true;                                 // the tile in one piece: its bucket counts are what hist_kernel found
  while (sb < b1) {                                  // one iteration unless the tile is overfull
    __syncthreads();
    if (t == 0) {
      u32 e = b1, n = tmp[29];
      if (sb != b0 || n > (u32)MAX_TILE) {             // overfull: as many consecutive blocks as the staging area takes
        e = sb; n = 0;
        while (e < b1 && n + rlen[e - b0] <= (u32)MAX_TILE) { n += rlen[e - b0]; e++; }   // (a single run never exceeds S W <= MAX_TILE)
        if (e == sb) { e = sb + 1; n = 0; }            // unreachable (static_assert above); never spin
      }
      tmp[30] = e; tmp[31] = n;
    }
    __syncthreads();
    const u32 se = tmp[30], cnt = tmp[31];
    whole = whole && sb == b0 && se == b1;
    if (whole) {
      // after colscan the histogram rows hold the exclusive prefix over the window's tiles: this tile's count of bucket b is the
      // next row's prefix (the bucket's total for the last tile) minus its own -- no second walk over the entries
      for (u32 b = t; b < nb; b += SORT_THREADS) lcnt[b] = (last_tile ? tot_w[b] : pre[nb + b]) - pre[b];
    } else {
      for (u32 b = t; b < nb; b += SORT_THREADS) lcnt[b] = 0;
      __syncthreads();
      for_each_run_entry(rlo, rlen, jobs.ent_off[job], SW, b0, sb, se, [&](int j, u64 at) { kk[j] = key[at]; }, [&](int j) { atomicAdd(&lcnt[kk[j]], 1u); });
    }
    __syncthreads();
    block_excl_scan2(lcnt, nb, tmp);
    for (u32 b = t; b < nb; b += SORT_THREADS) {
      const u32 here = (b + 1 < nb ? lcnt[b + 1] : cnt) - lcnt[b];
      cur[b] = lcnt[b]; gdst[b] = bs[b] + pre[b] + done[b] - lcnt[b]; done[b] += here;
    }
    __syncthreads();
    for_each_run_entry(rlo, rlen, jobs.ent_off[job], SW, b0, sb, se, [&](int j, u64 at) { kk[j] = key[at]; vv[j] = val[at]; },
                       [&](int j) {
                         const u32 pos = atomicAdd(&cur[kk[j]], 1u);
                         stage[pos] = vv[j];
                         sbkt[pos] = (unsigned short)kk[j];
                       });
    __syncthreads();
    for (u32 idx = t; idx < cnt; idx += SORT_THREADS) out[gdst[sbkt[idx]] + idx] = stage[idx];
    sb = se;
  }
}

// ---- bucket order: largest first -------------------------------------------------------------------------------
// A wave of the accumulate kernel runs as long as its largest bucket.  msm::accum_kernel sorts bucket sizes inside each
// block of 256; here all buckets of the launch are ordered by size with a counting sort (sizes are small integers), so
// the 64 lanes of every wave get equal trip counts (to within one), blocks start in longest-first order and the tail
// of the launch consists of the smallest buckets.
constexpr int SIZE_BINS = 1024;    // sizes >= 1023 share the last bin
// (Only the OWNED buckets are ordered -- slot o of the nj x nbown owned buckets is bucket owned_gid(o) --: a rank of 8 owns an eighth
// of every job's buckets, and ordering, permuting and launching the other seven eighths as empty buckets cost the size kernels
// their whole time and the accumulate kernel a tail of thousands of blocks that only find out that they have nothing to do.)
__device__ __forceinline__ u64 owned_gid(u64 o, u32 nbown, u32 nbt, u32 pshift, const Own& own) {
  const u64 k = o / nbown;
  const u32 ob = (u32)(o - k * nbown);
  const u32 v = own.first + (ob >> pshift) * own.stride;
  return k * nbt + ((u64)v << pshift) + (ob & ((1u << pshift) - 1));
}
__global__ __launch_bounds__(1024) void size_hist_kernel(const u32* __restrict__ tot, u64 OW, u32 nbown, u32 nbt, u32 pshift, Own own, u32* __restrict__ ghist) {
  __shared__ u32 lh[SIZE_BINS];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const u64 g = (u64)blockIdx.x * 1024 + threadIdx.x;
  if (g < OW) { u32 sz = tot[owned_gid(g, nbown, nbt, pshift, own)]; atomicAdd(&lh[sz < SIZE_BINS - 1 ? sz : SIZE_BINS - 1], 1u); }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], lh[threadIdx.x]);
}
// counts -> first position of each size in descending order
__global__ __launch_bounds__(1024) void size_scan_kernel(u32* __restrict__ ghist) {
  __shared__ u32 a[SIZE_BINS];
  __shared__ u32 tmp[32];
  a[threadIdx.x] = ghist[SIZE_BINS - 1 - threadIdx.x];
  __syncthreads();
  block_excl_scan2(a, SIZE_BINS, tmp);
  ghist[SIZE_BINS - 1 - threadIdx.x] = a[threadIdx.x];
}
// ---- accumulate over virtual slots: persistent waves, long buckets cut into PARTS (round 6) ------------------------------------
// Through round 5 the accumulate kernel was one thread per bucket in blocks of 256, dispatched by the hardware in size order.  Two
// things were lost that way.  (1) A bucket-range shard of G ranks has 1 / G of the buckets with lists as long as on one GPU: a rank
// of 8 has ~4 waves per SIMD in a launch, and the launch lasts as long as its LONGEST buckets (the mask polynomial's, 3 H
// coefficients: 102 entries where the launch's average is 37) while the other wave slots stand empty -- 63-75 % of the slots
// occupied over the three commit rounds of a proof at 2^20 (profiles/r06r_*).  (2) On one GPU a block's four waves end at
// different times and their slots wait for a whole new block.  Now the kernel is persistent -- two waves per SIMD (the group law
// with its multiplications interleaved fills a SIMD from two waves: x30.cuh), each taking the next 64 VIRTUAL slots from a counter
// until none is left -- and a bucket larger than Ts entries is accumulated in ceil(size / T) parts by as many lanes (T, Ts from the
// host: half and two thirds of a lane's share of the launch's entries, so on one GPU nothing is cut; doubled on the device until
// the parts fit the buffer that holds them); slots go in size order, largest first, so a wave's lanes run equally long, and a
// small kernel adds a bucket's parts up (one general addition per extra part).  Virtual slot -> (size class s, bucket r of the
// class, part j): the classes' first slots (voff) are searched in LDS; class s starts at pos[s] in `perm`.  The cut buckets are
// the largest, so their slots come first: slot = index into the parts' buffer.
struct VTab {
  u32 voff[SIZE_BINS + 1];   // [i]: first virtual slot of size class SIZE_BINS - 1 - i (descending sizes); [SIZE_BINS] = V
  u32 pos[SIZE_BINS + 1];    // [i]: first position in perm of that class; [SIZE_BINS] = owned buckets
  u32 V, T, Ts, nsplit, counter; // virtual slots, entries per part, largest uncut size, buckets with more than one part (= perm's first nsplit), next free slot
};
// parts of a bucket of size class s: one up to Ts entries, else ceil(s / T) -- Ts > T keeps the many buckets just above the part
// size whole (each extra part is one general addition in merge_parts_kernel and buys nothing there)
__device__ __forceinline__ u32 vparts(u32 s, u32 T, u32 Ts) { return s <= Ts ? 1u : (s + T - 1) / T; }
// one block of SIZE_BINS threads, after size_hist: ghist (counts) -> ghist (first positions, descending sizes: what size_perm
// counts up from) and the table above.  vcap: slots of the parts' buffer (the parts of the CUT buckets must fit).
__global__ __launch_bounds__(1024) void size_vscan_kernel(u32* __restrict__ ghist, VTab* __restrict__ vt, u32 T0, u32 Ts0, u32 vcap) {
  __shared__ u32 a[SIZE_BINS];
  __shared__ u32 b[SIZE_BINS];
  __shared__ u32 tmp[32];
  __shared__ u32 sT;
  const u32 i = threadIdx.x, s = SIZE_BINS - 1 - i;
  const u32 cnt = ghist[s];
  u32 T = T0 < 1 ? 1 : T0, Ts = Ts0 < T ? T : Ts0;
  for (;;) {                                  // a coarser cut until the parts fit (block-uniform loop; ends at the latest with Ts >= every class)
    b[i] = s > Ts ? cnt * vparts(s, T, Ts) : 0;
    __syncthreads();
    for (u32 off = SIZE_BINS / 2; off > 0; off >>= 1) { if (i < off) b[i] += b[i + off]; __syncthreads(); }
    if (i == 0) sT = b[0] <= vcap ? 1 : 0;
    __syncthreads();
    if (sT) break;
    T = T * 2; Ts = Ts < 0x40000000u ? Ts * 2 : Ts;
    __syncthreads();
  }
  a[i] = cnt; b[i] = cnt * vparts(s, T, Ts);
  __syncthreads();
  block_excl_scan2(a, SIZE_BINS, tmp);
  block_excl_scan2(b, SIZE_BINS, tmp);
  ghist[s] = a[i];
  vt->pos[i] = a[i]; vt->voff[i] = b[i];
  if (i == SIZE_BINS - 1) { vt->pos[SIZE_BINS] = a[i] + cnt; vt->voff[SIZE_BINS] = b[i] + cnt * vparts(s, T, Ts); vt->V = b[i] + cnt * vparts(s, T, Ts); vt->T = T; vt->Ts = Ts; vt->counter = 0; }
  // buckets with more than one part: the classes s > Ts -- the first position of class Ts in the descending order
  if (s == (Ts < SIZE_BINS - 1 ? Ts : SIZE_BINS - 1)) vt->nsplit = Ts < SIZE_BINS - 1 ? a[i] : 0;
}
// the last index i with tab[i] <= x (tab ascending, tab[0] = 0 <= x < tab[SIZE_BINS]): empty classes repeat their neighbour's value
__device__ __forceinline__ u32 vclass_of(const u32* tab, u32 x) {
  u32 lo = 0, hi = SIZE_BINS;                 // invariant: tab[lo] <= x < tab[hi]
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tab[mid] <= x) lo = mid; else hi = mid; }
  return lo;
}
__global__ __launch_bounds__(1024) void size_perm_kernel(const u32* __restrict__ tot, u64 OW, u32 nbown, u32 nbt, u32 pshift, Own own, u32* __restrict__ gcur,
                                                         u32* __restrict__ perm) {
  __shared__ u32 lh[SIZE_BINS];
  __shared__ u32 lbase[SIZE_BINS];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const u64 g = (u64)blockIdx.x * 1024 + threadIdx.x;
  const u64 gid = g < OW ? owned_gid(g, nbown, nbt, pshift, own) : 0;
  u32 bin = 0, rank = 0;
  if (g < OW) { u32 sz = tot[gid]; bin = sz < SIZE_BINS - 1 ? sz : SIZE_BINS - 1; rank = atomicAdd(&lh[bin], 1u); }
  __syncthreads();
  if (lh[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&gcur[threadIdx.x], lh[threadIdx.x]);
  __syncthreads();
  if (g < OW) perm[lbase[bin] + rank] = (u32)gid;
}

// ---- accumulate: thread per bucket, XYZZ += table point, all in 30-bit limbs -----------------------------------
// Same structure as msm::accum_kernel (first entry seeds the accumulator, collisions are deferred to the fix-up), buckets
// taken in the size order computed above, with the lazily reduced arithmetic of fq30.cuh.  Value bounds in units of
// p (a product of u and v is below 1 + u v / 630): table x < 1, +-y <= 2, zz, zzz <= 1.1; X1 <= 6.2, Y1 <= 2 are
// loop invariants: P = U2 - X1 + 8p <= 9.1, R = S2 - Y1 + 4p <= 5.1, PP, PPP, Q, R^2 <= 1.2,
// X3 = (R^2 - PPP + 2p) - 2Q + 3p <= 6.2, Y3 = (R (Q - X3 + 8p) + (4p - Y1) PPP) / R' <= 1 + (5.1 x 9.2 + 4 x 1.2) / 630 < 1.1.
// (Since round 6 this kernel is the cross-check, MH_ACC_PARTS=0; the default is accum30v_kernel below.)
template <int WAVES>
__global__ __launch_bounds__(msm::ACC_TPB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void accum30_kernel(const FbWin* __restrict__ fbw, const G1Aff30* __restrict__ table,
                                                               const u32* __restrict__ sorted_all, const u32* __restrict__ base,
                                                               const u32* __restrict__ tot, const u32* __restrict__ perm,
                                                               G1Xyzz30* __restrict__ buckets, u32* __restrict__ pend,
                                                               u32* __restrict__ n_deferred, u32 nb, u64 OW,
                                                               const u32* __restrict__ largest, u32 skew_limit) {
  const u64 slot = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= OW) return;                     // OW: the owned buckets of all jobs (perm lists exactly those, largest first)
  // a batch whose largest bucket exceeds the limit leaves this path (the host sees the same number with the results and takes
  // the variable-base path): return at once instead of walking a list of millions in one thread
  if (*largest > skew_limit) return;
  const u64 gid = perm[slot];
  const FbWin d = fbw[gid / nb];
  const u32* lst = sorted_all + d.off + base[gid];
  const G1Aff30* tab = table + d.delta;
  const u32 cnt = tot[gid];
  if (cnt == 0) { x30_store(buckets + gid, x30_identity()); pend[gid] = 0; return; }
  Fq30 zero;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) zero.v[i] = 0;
  Fq30 X1, Y1, ZZ, ZZZ;
  {
    const u32 e = lst[0];
    const G1Aff30* q = tab + (e & 0x7fffffffu);
    X1 = load30(q->x);
    Y1 = load30(q->y);
    if (e & 0x80000000u) Y1 = f30_sub<2>(zero, Y1);
#pragma unroll
    for (int i = 0; i < Fq30::NL; i++) { ZZ.v[i] = Fq30Params::ONE[i]; ZZZ.v[i] = Fq30Params::ONE[i]; }
  }
  u32 e_next = cnt > 1 ? lst[1] : 0;
  for (u32 k = 1; k < cnt; k++) {
    const u32 e = e_next;
    if (k + 1 < cnt) e_next = lst[k + 1];           // one iteration ahead: takes the list load off the critical path
    const G1Aff30* q = tab + (e & 0x7fffffffu);
    const Fq30 x2 = load30(q->x);
    Fq30 y2 = load30(q->y);
    if (e & 0x80000000u) y2 = f30_sub<2>(zero, y2);
    const Fq30 P = f30_sub<8>(f30_mul(x2, ZZ), X1);
    // equal x (the same point twice, or P and -P): leave the whole bucket to the fix-up pass, which recomputes it with the
    // complete addition law.  The lists are read-only here because two jobs may share them (FbWin::delta).
    if (__builtin_expect(f30_is_zero(P), 0)) { pend[gid] = 1; atomicAdd(n_deferred, 1u); return; }
    const Fq30 R = f30_sub<4>(f30_mul(y2, ZZZ), Y1);
    Fq30 PP = f30_sqr(P);
    ZZ = f30_mul(ZZ, PP);
    const Fq30 Q = f30_mul(X1, PP);
    PP = f30_mul(P, PP);                 // PPP
    ZZZ = f30_mul(ZZZ, PP);
    const Fq30 nY1 = f30_sub<4>(zero, Y1);                      // 4p - Y1
    X1 = f30_sub2<3>(f30_sub<2>(f30_sqr(R), PP), Q);
    Y1 = f30_mul2(R, f30_sub<8>(Q, X1), nY1, PP);               // Y3 = R (Q - X3) - Y1 PPP: two products, ONE reduction
  }
  store30(buckets[gid].c[0], X1); store30(buckets[gid].c[1], Y1); store30(buckets[gid].c[2], ZZ); store30(buckets[gid].c[3], ZZZ);
  pend[gid] = 0;
}

// The same accumulation over VIRTUAL slots (see VTab): persistent waves, a bucket of more than T entries in parts.  A part's sum goes
// to aux[slot] (merge_parts_kernel adds a bucket's parts up), an uncut bucket's straight to `buckets`.  `pend` is zeroed by the host
// beforehand -- a part that meets an equal-x pair marks the whole bucket for the fix-up, the other parts must not unmark it.
__global__ __launch_bounds__(msm::ACC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void accum30v_kernel(
    const FbWin* __restrict__ fbw, const G1Aff30* __restrict__ table, const u32* __restrict__ sorted_all, const u32* __restrict__ base,
    const u32* __restrict__ tot, const u32* __restrict__ perm, G1Xyzz30* __restrict__ buckets, G1Xyzz30* __restrict__ aux, u32* __restrict__ pend,
    u32* __restrict__ n_deferred, u32 nb, VTab* __restrict__ vt, const u32* __restrict__ largest, u32 skew_limit) {
  __shared__ u32 s_voff[SIZE_BINS + 1];
  __shared__ u32 s_pos[SIZE_BINS + 1];
  if (*largest > skew_limit) return;
  for (u32 i = threadIdx.x; i <= SIZE_BINS; i += blockDim.x) { s_voff[i] = vt->voff[i]; s_pos[i] = vt->pos[i]; }
  __syncthreads();
  const u32 V = vt->V, T = vt->T, Ts = vt->Ts, lane = threadIdx.x & 63u;
  Fq30 zero;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) zero.v[i] = 0;
  for (;;) {
    u32 first = 0;
    if (lane == 0) first = atomicAdd(&vt->counter, 64u);
    first = (u32)__shfl((int)first, 0);
    if (first >= V) break;
    const u32 slot = first + lane;
    if (slot >= V) continue;
    const u32 ci = vclass_of(s_voff, slot);
    const u32 p = vparts(SIZE_BINS - 1 - ci, T, Ts);
    const u32 rel = slot - s_voff[ci], r = rel / p, j = rel - r * p;
    const u32 gid = perm[s_pos[ci] + r];
    const FbWin d = fbw[gid / nb];
    const u32 cnt = tot[gid];
    const u32 lo = (u32)((u64)cnt * j / p), hi = (u32)((u64)cnt * (j + 1) / p);
    G1Xyzz30* dst = p == 1 ? buckets + gid : aux + slot;
    if (hi == lo) { x30_store(dst, x30_identity()); continue; }
    const u32* lst = sorted_all + d.off + base[gid];
    const G1Aff30* tab = table + d.delta;
    Fq30 X1, Y1, ZZ, ZZZ;
    {
      const u32 e = lst[lo];
      const G1Aff30* q = tab + (e & 0x7fffffffu);
      X1 = load30(q->x);
      Y1 = load30(q->y);
      if (e & 0x80000000u) Y1 = f30_sub<2>(zero, Y1);
#pragma unroll
      for (int i = 0; i < Fq30::NL; i++) { ZZ.v[i] = Fq30Params::ONE[i]; ZZZ.v[i] = Fq30Params::ONE[i]; }
    }
    u32 e_next = lo + 1 < hi ? lst[lo + 1] : 0;
    bool deferred = false;
    for (u32 k = lo + 1; k < hi; k++) {
      const u32 e = e_next;
      if (k + 1 < hi) e_next = lst[k + 1];
      const G1Aff30* q = tab + (e & 0x7fffffffu);
      const Fq30 x2 = load30(q->x);
      Fq30 y2 = load30(q->y);
      if (e & 0x80000000u) y2 = f30_sub<2>(zero, y2);
      // accum30_kernel's group law (value bounds there) with its ten multiplications as four groups of interleaved chains -- {U2, S2},
      // {PP, R^2}, {Q, PPP, ZZ3}, {Y3, ZZZ3}: fq30.cuh f30_mul_x2 ... -- so that two resident waves fill the SIMD
      Fq30 U2, S2;
      f30_mul_x2(U2, x2, ZZ, S2, y2, ZZZ);
      const Fq30 P = f30_sub<8>(U2, X1);
      if (__builtin_expect(f30_is_zero(P), 0)) { deferred = true; break; }
      const Fq30 R = f30_sub<4>(S2, Y1);
      Fq30 PP, RR, Q, PPP;
      f30_sqr_x2(PP, P, RR, R);
      f30_mul_x3(Q, X1, PP, PPP, P, PP, ZZ, ZZ, PP);
      const Fq30 nY1 = f30_sub<4>(zero, Y1);
      X1 = f30_sub2<3>(f30_sub<2>(RR, PPP), Q);
      f30_mul2_mul(Y1, R, f30_sub<8>(Q, X1), nY1, PPP, ZZZ, ZZZ, PPP);
    }
    if (deferred) { pend[gid] = 1; atomicAdd(n_deferred, 1u); continue; }     // the fix-up recomputes the WHOLE bucket
    store30(dst->c[0], X1); store30(dst->c[1], Y1); store30(dst->c[2], ZZ); store30(dst->c[3], ZZZ);
  }
}
// the buckets that were accumulated in parts (perm's first nsplit): bucket = sum of its parts.  (A bucket marked for the fix-up has
// parts nobody wrote; what is formed of them here is overwritten by the fix-up, which runs after this.)
__global__ __launch_bounds__(128) void merge_parts_kernel(const VTab* __restrict__ vt, const u32* __restrict__ perm, const G1Xyzz30* __restrict__ aux,
                                                          G1Xyzz30* __restrict__ buckets, const u32* __restrict__ largest, u32 skew_limit) {
  if (*largest > skew_limit) return;
  const u32 n = vt->nsplit, T = vt->T, Ts = vt->Ts;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 ci = vclass_of(vt->pos, i);
    const u32 p = vparts(SIZE_BINS - 1 - ci, T, Ts);
    const G1Xyzz30* part = aux + vt->voff[ci] + (u64)(i - vt->pos[ci]) * p;
    X30 acc = x30_load(part);
    for (u32 j = 1; j < p; j++) { const X30 b = x30_load(part + j); x30_add_ilp_inl(acc, b); }
    x30_store(buckets + perm[i], acc);
  }
}

// buckets that met an equal-x pair (pend != 0): recomputed from their whole list with the complete group law in the
// standard representation, one thread per such bucket (a small grid-stride launch that returns at once in the usual case
// of no deferred bucket at all)
__global__ __launch_bounds__(64) void fixup30_kernel(const FbWin* __restrict__ fbw, const G1Aff30* __restrict__ table,
                                                     const u32* __restrict__ sorted_all, const u32* __restrict__ base,
                                                     const u32* __restrict__ tot, u32* __restrict__ pend,
                                                     const u32* __restrict__ n_deferred, G1Xyzz30* __restrict__ buckets, u32 nb, u64 WB) {
  if (*n_deferred == 0) return;
  for (u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x; gid < WB; gid += (u64)gridDim.x * blockDim.x) {
  if (pend[gid] == 0) continue;
  const FbWin d = fbw[gid / nb];
  const u32* lst = sorted_all + d.off + base[gid];
  const G1Aff30* tab = table + d.delta;
  const u32 cnt = tot[gid];
  G1Xyzz acc = G1Xyzz::identity();
  for (u32 k = 0; k < cnt; k++) {
    const u32 e = lst[k];
    const G1Aff30* q = tab + (e & 0x7fffffffu);
    Fq x = f30_to_fq(load30(q->x)), y = f30_to_fq(load30(q->y));
    if (e & 0x80000000u) y = ff_neg(y);
    g1_madd(acc, x, y);
  }
  x30_store(buckets + gid, x30_from_std(acc));
  pend[gid] = 0;                                  // settled (MH_CHECK counts what is still pending afterwards)
  }
}

// ---- bucket reduction: row / column sums of the bucket matrix, then bit-plane sums of those ------------------------------
// sum_b (b + 1) B_b over the job's single bucket set (b = 0-based bucket index across the virtual windows).  Rounds 1-4 ran the
// classical form -- a thread per segment of `seg` buckets: 2 seg running-sum additions plus (segment offset) x (segment total)
// by a ~19-bit double-and-add, then a tree over the segment results -- a chain of 2 seg + ~25 + ~14 dependent group
// operations of ~15 us each at one wave per SIMD (profiles/r04o_last_prove_kernels_2p20.txt: 6.83 ms per proof at 2^20), a
// third of it the offset multiplication that every thread repeats.  Here no point is ever multiplied on the device:
//   * the owned buckets of a job are a matrix S[m][c] (m < R_own rows of C = 2^lgC buckets; a partition is rpp whole rows, so
//     a rank's rows are its partitions' rows), bucket index b = r(m) C + c with r(m) = (first + (m / rpp) stride) rpp + m % rpp;
//   * rsum_kernel: Row_m = sum_c S[m][c] and Col_c = sum_m S[m][c].  A group of J (rows) or I (columns) adjacent lanes shares
//     one sum: every lane adds its strided share of <= Lr or Lc buckets, then the group's lanes are summed by an xor butterfly
//     of lg J (lg I) shuffled additions -- 2 additions per bucket in all, a chain of L + lg G per thread, any number of
//     threads (two waves per SIMD: the loop has ONE inlined copy of the group law, 220 registers);
//   * plane_kernel: sum_m m Row_m and sum_c c Col_c as BIT PLANES -- Y_p = sum of the Col_c whose c has bit p, M_p likewise
//     over the bits of m, T = sum of all rows: one block per (job, plane), members enumerated directly from the bit layout, a
//     tree through LDS -- lg(owned buckets) + 1 points per job go to the host;
//   * the host (capi.hip: FbRun::finish) combines them with the coefficients 2^p (Y_p), C 2^p or C stride 2^p (M_p: r(m) =
//     first rpp + stride rpp (m / rpp) + m % rpp) and C first rpp + 1 (T): one shared chain of ~20 doublings on a CPU core,
//     ~50 us per job, jobs side by side on the host pool.
// Depth per batch: L + lg G + 2 + 8 additions instead of 2 seg + 25 + 14, and 2 additions per bucket instead of ~2.9.
// (First form of round 5, profiles/r05a_*: plane sums over the per-thread partials instead of over whole rows and columns --
// 16 x 20 x 4 blocks of LDS trees per batch, 1.0 ms per launch: slower than what it replaced.)
struct RsPlan {
  u32 lgC, C;        // columns per row (C divides the partition size)
  u32 lgrpp;         // lg(rows per partition)
  u32 R_own;         // owned rows
  u32 lgJ, J, Lr;    // lanes per row sum, buckets per lane (J Lr = C)
  u32 lgI, I, Lc;    // lanes per column sum, buckets per lane (I Lc >= R_own)
  u32 NTr, NT;       // threads of the row sums (R_own J rounded up to whole waves), all threads per job (+ C I, likewise)
  u32 NS;            // sums per job: R_own rows, then C columns
  u32 lgM;           // bits of a row index m < R_own
  u32 nplanes;       // lgC column planes, lgM row planes, the total
};
constexpr int PLANE_THREADS = 256;
__device__ __forceinline__ u32 insert_one(u32 x, u32 p) { return ((x >> p) << (p + 1)) | (1u << p) | (x & ((1u << p) - 1)); }
__device__ __forceinline__ Fq30 f30_shfl_xor(const Fq30& a, u32 mask) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) r.v[i] = (u32)__shfl_xor((int)a.v[i], (int)mask);
  return r;
}

// thread q of job w: q < NTr: lane g = q % J of row m = q / J adds columns g, g + J, ...; else lane g = (q - NTr) % I of column
// c = (q - NTr) / I adds rows g, g + I, ...  NTr and NT are whole waves, so a wave is all rows or all columns: its trip count
// and its shuffles are uniform (threads past the last row / column carry the identity through them).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rsum_kernel(
    const G1Xyzz30* __restrict__ buckets, G1Xyzz30* __restrict__ sums, u32 nbt, u32 njobs, RsPlan p, Own own,
    const u32* __restrict__ largest, u32 skew_limit) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= njobs * p.NT) return;            // whole waves only (NT is a multiple of 64)
  if (*largest > skew_limit) return;        // a skewed batch left this path in the accumulate kernel: the buckets hold nothing
  const u32 w = t / p.NT, q = t % p.NT;
  const G1Xyzz30* B = buckets + (u64)w * nbt;
  const bool row = q < p.NTr;
  u32 grp, g, lgG, L, m, c, dm, dc;
  bool valid;
  if (row) { grp = q >> p.lgJ; g = q & (p.J - 1); lgG = p.lgJ; L = p.Lr; m = grp; c = g; dm = 0; dc = p.J; valid = grp < p.R_own; }
  else { const u32 q2 = q - p.NTr; grp = q2 >> p.lgI; g = q2 & (p.I - 1); lgG = p.lgI; L = p.Lc; m = g; c = grp; dm = p.I; dc = 0; valid = grp < p.C; }
  X30 acc = x30_identity();
  for (u32 step = 0; step < L + lgG; step++) {
    X30 b;
    if (step < L) {
      if (valid && m < p.R_own && c < p.C) {
        const u32 v = own.first + (m >> p.lgrpp) * own.stride;
        const u32 r = (v << p.lgrpp) | (m & ((1u << p.lgrpp) - 1));
        b = x30_load(B + (((u64)r << p.lgC) | c));
      } else b = x30_identity();
      m += dm; c += dc;
    } else {
      const u32 mask = 1u << (step - L);
      b.x = f30_shfl_xor(acc.x, mask); b.y = f30_shfl_xor(acc.y, mask); b.zz = f30_shfl_xor(acc.zz, mask); b.zzz = f30_shfl_xor(acc.zzz, mask);
    }
    x30_add_ilp_inl(acc, b);
  }
  if (valid && g == 0) x30_store(sums + (u64)w * p.NS + (row ? grp : p.R_own + grp), acc);
}

// tree sum of the block's first `nthreads` accumulators (a power of two) through LDS; the result is thread 0's `acc`
__device__ __forceinline__ void block_tree_sum(X30& acc, G1Xyzz30* sh, u32 nthreads) {
  if (threadIdx.x < nthreads) x30_store(sh + threadIdx.x, acc);
  __syncthreads();
  for (u32 off = nthreads >> 1; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      X30 a = x30_load(sh + threadIdx.x);
      const X30 b = x30_load(sh + threadIdx.x + off);
      x30_add_ilp_inl(a, b);
      if (off > 1) x30_store(sh + threadIdx.x, a); else acc = a;
    }
    __syncthreads();
  }
}

// grid (nplanes, njobs): block (plane, w) sums the members of one plane of job w -> out_std[w nplanes + plane], standard representation
__global__ __launch_bounds__(PLANE_THREADS) void plane_kernel(const G1Xyzz30* __restrict__ sums, G1Xyzz* __restrict__ out_std, RsPlan p,
                                                              const u32* __restrict__ largest, u32 skew_limit) {
  extern __shared__ __attribute__((aligned(16))) u32 lds_plane[];
  G1Xyzz30* sh = reinterpret_cast<G1Xyzz30*>(lds_plane);
  if (*largest > skew_limit) return;
  const u32 plane = blockIdx.x, w = blockIdx.y;
  const G1Xyzz30* S = sums + (u64)w * p.NS;
  u32 count;                                               // members enumerated below (some row indices fall past R_own)
  if (plane < p.lgC) count = p.C >> 1;
  else if (plane < p.lgC + p.lgM) count = 1u << (p.lgM - 1);
  else count = p.R_own;
  X30 acc = x30_identity();
  for (u32 k = threadIdx.x; k < count; k += PLANE_THREADS) {
    u32 q;
    if (plane < p.lgC) q = p.R_own + insert_one(k, plane);                 // columns c with bit `plane`
    else if (plane < p.lgC + p.lgM) { q = insert_one(k, plane - p.lgC); if (q >= p.R_own) continue; }   // rows m with bit (plane - lgC)
    else q = k;                                                            // every row: the rows cover every bucket once
    const X30 t = x30_load(S + q);
    x30_add_ilp_inl(acc, t);
  }
  u32 width = 1;
  while (width < count && width < PLANE_THREADS) width <<= 1;
  if (width > 1) block_tree_sum(acc, sh, width);
  if (threadIdx.x == 0) g1_store_xyzz(out_std + (u64)w * p.nplanes + plane, x30_to_std(acc));
}

}  // namespace msmfb

