// Invariants of the fixed-base MSM pipeline (msm_fb.cuh), checked per batch when MH_CHECK >= 1 (mh_check_level): every stage's
// output is compared with what its INPUT says it must be, so that a wrong group element -- VariableBaseMSM::multi_scalar_mul
// under /root/reference src/lib.rs:172,193,213,292 has exactly one right answer -- names the stage that produced it instead of
// "proof differs".  Everything here reads; nothing the product path computes depends on it.
//
//   level 1 (cheap, a few ms per batch at 2^20):
//     recode   : count / sum / xor of the entries every virtual window MUST hold, straight from the scalars
//     entries  : the same three numbers of what `psplit` wrote (val, run by run; the run tables well-formed) and of what `scatter`
//                wrote (sorted): sorted is a permutation of val, val is what the scalars say; the count is also what the device
//                put into the window's descriptor (woff_kernel)
//     tot/base : sum of a window's bucket sizes = its entries; bucket starts = exclusive scan of the sizes
//     buckets  : every owned bucket after accumulate + fix-up is the identity or a point of the curve (the buffers are filled
//                with 0xA5 bytes before the accumulation, so a bucket nothing wrote is caught), no bucket left pending
//     sums     : every row / column sum and every plane is on the curve; sum of the rows = sum of the columns = the T plane
//   level 2 (a second, independent computation; sized for tests and the soak tool):
//     lists    : every entry of every bucket's list recodes to that bucket and sign
//     buckets  : every bucket recomputed from its list with the 32-bit complete law and compared as a group element
//     result   : sum (b + 1) B_b by running sums on the host from the downloaded buckets (small bucket sets only)
#pragma once
#include "msm_fb.cuh"

namespace msmchk {
using msmfb::FbJobs; using msmfb::FbWin; using msmfb::Own; using msmfb::G1Xyzz30; using msmfb::G1Aff30; using msmfb::X30;
using msm::Windows;

struct Sum { unsigned long long cnt, sum; u32 x, pad; };

__device__ __forceinline__ void wave_flush(Sum* dst, unsigned long long cnt, unsigned long long sum, u32 x) {
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_down(cnt, off); sum += __shfl_down(sum, off); x ^= (u32)__shfl_down((int)x, off);
  }
  if ((threadIdx.x & 63) == 0 && cnt) { atomicAdd(&dst->cnt, cnt); atomicAdd(&dst->sum, sum); atomicXor(&dst->x, x); }
}

// grid (blocks, njobs): what the entry lists of job's windows must hold
__global__ __launch_bounds__(256) void recode_kernel(FbJobs jobs, Sum* __restrict__ out, u32 W, Windows win, int is_mont, u32 nparts,
                                                     u32 pshift, u32 tab_n, Own own) {
  __shared__ unsigned long long cnt[msmfb::MAX_PARTS], sum[msmfb::MAX_PARTS];
  __shared__ u32 xr[msmfb::MAX_PARTS];
  const u32 job = blockIdx.y;
  if (jobs.nblk[job] == 0) return;                 // shares another job's lists (FbWin::delta)
  for (u32 v = threadIdx.x; v < msmfb::MAX_PARTS; v += blockDim.x) { cnt[v] = 0; sum[v] = 0; xr[v] = 0; }
  __syncthreads();
  const u64 n = jobs.n[job];
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    Fr s = ff_load(jobs.scalars[job] + i);
    if (!is_mont) s = ff_to_mont(s);
    const u32 t0 = jobs.tab_off[job] + (u32)i * jobs.tab_stride[job];
    msm::for_each_digit(s, W, win, [&](u32 w, u32 e) {
      u32 b = e & 0x7fffffffu;
      if (!b) return;
      const u32 v = (b - 1) >> pshift;
      if (!msmfb::owns(own, v)) return;
      const u32 val = (w * tab_n + t0) | (e & 0x80000000u);
      atomicAdd(&cnt[v], 1ull); atomicAdd(&sum[v], (unsigned long long)val); atomicXor(&xr[v], val);
    });
  }
  __syncthreads();
  for (u32 v = threadIdx.x; v < nparts; v += blockDim.x)
    if (cnt[v]) { Sum* d = out + (u64)job * nparts + v; atomicAdd(&d->cnt, cnt[v]); atomicAdd(&d->sum, sum[v]); atomicXor(&d->x, xr[v]); }
}

// grid (max blocks, njobs): what the split wrote, block by block: the entries of split block b of a job lie at [b S W, ...) grouped by
// partition, lst holds the nparts + 1 run boundaries
__global__ __launch_bounds__(256) void runs_kernel(FbJobs jobs, const u32* __restrict__ val, const unsigned short* __restrict__ lst_all,
                                                   Sum* __restrict__ out, u32 nparts, u32 SW, u32* __restrict__ bad) {
  __shared__ unsigned long long cnt[msmfb::MAX_PARTS], sum[msmfb::MAX_PARTS];
  __shared__ u32 xr[msmfb::MAX_PARTS];
  const u32 job = blockIdx.y, blk = blockIdx.x;
  if (blk >= jobs.nblk[job]) return;
  for (u32 v = threadIdx.x; v < msmfb::MAX_PARTS; v += blockDim.x) { cnt[v] = 0; sum[v] = 0; xr[v] = 0; }
  __syncthreads();
  const unsigned short* L = lst_all + jobs.lst_off[job] + (u64)blk * (nparts + 1);
  const u32 total = L[nparts];
  if (threadIdx.x == 0) {                          // boundaries non-decreasing, the block's region not exceeded
    bool ok = L[0] == 0 && total <= SW;
    for (u32 v = 0; v < nparts; v++) ok = ok && L[v] <= L[v + 1];
    if (!ok) atomicAdd(bad, 1u);
  }
  const u32* a = val + jobs.ent_off[job] + (u64)blk * SW;
  for (u32 i = threadIdx.x; i < total && total <= SW; i += blockDim.x) {
    u32 lo = 0, hi = nparts;                        // the partition whose run holds entry i: L[lo] <= i < L[lo + 1]
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (L[mid] <= i) lo = mid; else hi = mid; }
    const u32 e = a[i];
    atomicAdd(&cnt[lo], 1ull); atomicAdd(&sum[lo], (unsigned long long)e); atomicXor(&xr[lo], e);
  }
  __syncthreads();
  for (u32 v = threadIdx.x; v < nparts; v += blockDim.x)
    if (cnt[v]) { Sum* d = out + (u64)job * nparts + v; atomicAdd(&d->cnt, cnt[v]); atomicAdd(&d->sum, sum[v]); atomicXor(&d->x, xr[v]); }
}

// grid (blocks, WT): what an entry array holds in window gw's run
__global__ __launch_bounds__(256) void entries_kernel(const FbWin* __restrict__ fbw, const u32* __restrict__ arr, Sum* __restrict__ out) {
  const u32 gw = blockIdx.y;
  const FbWin d = fbw[gw];
  if (d.ntiles == 0) return;                       // empty, or the lists of another job
  const u32* a = arr + d.off;
  unsigned long long cnt = 0, sum = 0; u32 x = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < d.cnt; i += gridDim.x * blockDim.x) { const u32 v = a[i]; cnt++; sum += v; x ^= v; }
  wave_flush(out + gw, cnt, sum, x);
}

// one wave per window: out[2 gw] = sum of the bucket sizes, out[2 gw + 1] = buckets whose start is not the exclusive scan
__global__ __launch_bounds__(64) void tot_kernel(const u32* __restrict__ tot, const u32* __restrict__ base, u32 nb, u32* __restrict__ out) {
  const u32 gw = blockIdx.x, l = threadIdx.x;
  const u32 per = (nb + 63) / 64, lo = l * per;
  u32 s = 0;
  for (u32 k = 0; k < per; k++) if (lo + k < nb) s += tot[(u64)gw * nb + lo + k];
  u32 inc = s;
  for (int off = 1; off < 64; off <<= 1) { const u32 o = __shfl_up(inc, off); if (l >= (u32)off) inc += o; }
  u32 run = inc - s, bad = 0;
  for (u32 k = 0; k < per; k++) if (lo + k < nb) { if (base[(u64)gw * nb + lo + k] != run) bad++; run += tot[(u64)gw * nb + lo + k]; }
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off);
  const u32 total = __shfl(inc, 63);
  if (l == 0) { out[2 * gw] = total; out[2 * gw + 1] = bad; }
}

// bad[0] = entries that do not belong where they are, bad[1] = bucket of the first one seen, bad[2] = that entry
__global__ __launch_bounds__(128) void lists_kernel(const FbWin* __restrict__ fbw, FbJobs jobs, const u32* __restrict__ sorted_all,
                                                    const u32* __restrict__ base, const u32* __restrict__ tot, u32 nb, u64 WB, u32 nparts,
                                                    u32 pshift, u32 W, Windows win, int is_mont, u32 tab_n, u32* __restrict__ bad) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= WB) return;
  const u32 gw = (u32)(gid / nb), job = gw / nparts, v = gw % nparts;
  if (jobs.nblk[job] == 0) return;
  const FbWin d = fbw[gw];
  const u32* lst = sorted_all + d.off + base[gid];
  const u32 cnt = tot[gid];
  const u32 bucket = (v << pshift) + (u32)(gid % nb) + 1;       // |digit| of every entry of this list
  for (u32 k = 0; k < cnt; k++) {
    const u32 e = lst[k], idx = e & 0x7fffffffu;
    const u32 w = idx / tab_n, t0 = idx % tab_n;
    bool ok = w < W && t0 >= jobs.tab_off[job] && (t0 - jobs.tab_off[job]) % jobs.tab_stride[job] == 0;
    const u64 i = ok ? (t0 - jobs.tab_off[job]) / jobs.tab_stride[job] : 0;
    ok = ok && i < jobs.n[job];
    if (ok) {
      Fr s = ff_load(jobs.scalars[job] + i);
      if (!is_mont) s = ff_to_mont(s);
      u32 want = 0;
      msm::for_each_digit(s, W, win, [&](u32 ww, u32 ee) { if (ww == w) want = ee; });
      ok = want == (bucket | (e & 0x80000000u));
    }
    if (!ok && atomicAdd(&bad[0], 1u) == 0) { bad[1] = (u32)gid; bad[2] = e; }
  }
}

__device__ __forceinline__ bool std_on_curve(const G1Xyzz& a) {
  if (a.is_identity()) return true;
  const Fq zzz2 = ff_sqr(a.zzz);
  const Fq rhs = ff_add(ff_mul(ff_sqr(a.x), a.x), ff_mul(G1_CURVE_B_MONT(), zzz2));
  return ff_sqr(a.y) == rhs && ff_mul(ff_sqr(a.zz), a.zz) == zzz2;
}
__device__ __forceinline__ bool std_same_point(const G1Xyzz& a, const G1Xyzz& b) {
  if (a.is_identity() || b.is_identity()) return a.is_identity() && b.is_identity();
  return ff_mul(a.x, b.zz) == ff_mul(b.x, a.zz) && ff_mul(a.y, b.zzz) == ff_mul(b.y, a.zzz);
}

// bad[0] = owned buckets that are neither the identity nor on the curve, bad[1] = the first one, bad[2] = owned buckets still pending;
// recompute != 0: also bad[3] = buckets that are not the sum of their list (32-bit complete law), bad[4] = the first one
__global__ __launch_bounds__(64) void buckets_kernel(const FbWin* __restrict__ fbw, const G1Aff30* __restrict__ table,
                                                     const u32* __restrict__ sorted_all, const u32* __restrict__ base, const u32* __restrict__ tot,
                                                     const G1Xyzz30* __restrict__ buckets, const u32* __restrict__ pend, u32 nb, u64 WB,
                                                     u32 nparts, Own own, int recompute, u32* __restrict__ bad) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= WB) return;
  const u32 gw = (u32)(gid / nb);
  if (!msmfb::owns(own, gw % nparts)) return;
  const G1Xyzz got = msmfb::x30_to_std(msmfb::x30_load(buckets + gid));
  if (!std_on_curve(got) && atomicAdd(&bad[0], 1u) == 0) bad[1] = (u32)gid;
  if (pend[gid] != 0) atomicAdd(&bad[2], 1u);
  if (!recompute) return;
  const FbWin d = fbw[gw];
  const u32* lst = sorted_all + d.off + base[gid];
  const G1Aff30* tab = table + d.delta;
  const u32 cnt = tot[gid];
  G1Xyzz acc = G1Xyzz::identity();
  for (u32 k = 0; k < cnt; k++) {
    const u32 e = lst[k];
    const G1Aff30* q = tab + (e & 0x7fffffffu);
    Fq x = f30_to_fq(msmfb::load30(q->x)), y = f30_to_fq(msmfb::load30(q->y));
    if (e & 0x80000000u) y = ff_neg(y);
    g1_madd(acc, x, y);
  }
  if (!std_same_point(acc, got) && atomicAdd(&bad[3], 1u) == 0) bad[4] = (u32)gid;
}

// n lazily reduced points (row / column sums): bad[0] = off the curve, bad[1] = the first
__global__ __launch_bounds__(64) void sums_kernel(const G1Xyzz30* __restrict__ pts, u64 n, u32* __restrict__ bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!std_on_curve(msmfb::x30_to_std(msmfb::x30_load(pts + i))) && atomicAdd(&bad[0], 1u) == 0) bad[1] = (u32)i;
}
__global__ __launch_bounds__(64) void planes_kernel(const G1Xyzz* __restrict__ pts, u64 n, u32* __restrict__ bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!std_on_curve(g1_load_xyzz(pts + i)) && atomicAdd(&bad[0], 1u) == 0) bad[1] = (u32)i;
}

// grid (2, njobs): block (0, w) = sum of job w's row sums, (1, w) = sum of its column sums -> out[2 w + which], standard form
__global__ __launch_bounds__(msmfb::PLANE_THREADS) void rowcol_kernel(const G1Xyzz30* __restrict__ sums, G1Xyzz* __restrict__ out, msmfb::RsPlan p) {
  extern __shared__ __attribute__((aligned(16))) u32 lds_chk[];
  G1Xyzz30* sh = reinterpret_cast<G1Xyzz30*>(lds_chk);
  const u32 which = blockIdx.x, w = blockIdx.y;
  const G1Xyzz30* S = sums + (u64)w * p.NS + (which ? p.R_own : 0);
  const u32 count = which ? p.C : p.R_own;
  X30 acc = msmfb::x30_identity();
  for (u32 k = threadIdx.x; k < count; k += msmfb::PLANE_THREADS) { const X30 t = msmfb::x30_load(S + k); msmfb::x30_add_ilp(acc, t); }
  msmfb::block_tree_sum(acc, sh, msmfb::PLANE_THREADS);
  if (threadIdx.x == 0) g1_store_xyzz(out + 2 * w + which, msmfb::x30_to_std(acc));
}

// n lazily reduced points -> the standard representation (for the host's own recomputation of a result)
__global__ __launch_bounds__(64) void to_std_kernel(const G1Xyzz30* __restrict__ in, G1Xyzz* __restrict__ out, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g1_store_xyzz(out + i, msmfb::x30_to_std(msmfb::x30_load(in + i)));
}

}  // namespace msmchk
