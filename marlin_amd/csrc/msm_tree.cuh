// Bucket accumulation by a pairwise tree of AFFINE additions with grid-wide batched inversion.
//
// Alternative to msm::accum_kernel (XYZZ accumulators, 8M+2S = 10 Fq multiplications per point):
// an affine addition costs lambda = (y2-y1)/(x2-x1), x3 = lambda^2-x1-x2, y3 = lambda(x1-x3)-y1, i.e. one
// inversion + 2M + 1S; Montgomery's trick turns the inversions of a whole batch into 3 multiplications
// each, so an addition costs 6 multiplications -- but only if the additions of a batch are independent.
// Summing a bucket sequentially is not; summing it as a binary tree is: in round r every bucket's
// m_r points are added in disjoint pairs (an odd one is carried), giving m_{r+1} = ceil(m_r / 2) points.
// All pairs of all buckets of all windows form one batch per round:
//     fwd  : thread t walks its outputs o = t, t+T, t+2T, ...; for each it finds the owning bucket
//            (binary search, stored in ob[o]), classifies the pair, multiplies the running product of the
//            denominators d_o and stores the prefix in pre[o]; the thread's total goes to prod[t].
//     inv  : prod[0..T) is inverted with Montgomery's trick again (32 per thread, one Fermat inversion each):
//            ~T/32 inversions for the whole round instead of one per thread.
//     bwd  : thread t unwinds its chain from the end: 1/d_o = inv * pre[o], inv *= d_o, completes the
//            addition and writes the affine result to out[o].
// Outputs are strided by T across a thread's chain so that a wave's lanes touch consecutive outputs
// (coalesced 96-B points, prefixes and owner ids).  A bucket holding all n points (every scalar equal)
// takes log2 n rounds of fully parallel work instead of n sequential additions.
// Doubling (P1 == P2), cancellation (P1 == -P2) and points at infinity are handled in place; the point at
// infinity is stored as (0, 0), which is not on y^2 = x^3 + 4.
#pragma once
#include "g1.cuh"
#include "msm.cuh"

namespace msmtree {

constexpr int TPB = 256;

// ---- exclusive scan of u32 (n up to 2^32), out[n] = total ------------------------------------------------
__global__ __launch_bounds__(256) void scan_reduce_kernel(const u32* __restrict__ in, u32* __restrict__ sums, u64 n) {
  __shared__ u32 sh[256];
  u64 base = (u64)blockIdx.x * 1024;
  u32 s = 0;
  for (int k = 0; k < 4; k++) { u64 i = base + threadIdx.x * 4 + k; if (i < n) s += in[i]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(u32* __restrict__ sums, u64 nblocks, u32* __restrict__ total) {
  __shared__ u32 part[1024];
  u64 per = (nblocks + 1023) / 1024;
  u64 lo = threadIdx.x * per;
  u32 s = 0;
  for (u64 k = 0; k < per; k++) if (lo + k < nblocks) s += sums[lo + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - s;
  for (u64 k = 0; k < per; k++) if (lo + k < nblocks) { u32 v = sums[lo + k]; sums[lo + k] = run; run += v; }
  if (threadIdx.x == 1023) *total = part[1023];
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const u32* __restrict__ in, const u32* __restrict__ sums,
                                                         u32* __restrict__ out, u64 n, const u32* __restrict__ total) {
  __shared__ u32 sh[256];
  u64 base = (u64)blockIdx.x * 1024;
  u32 f[4]; u32 s = 0;
  for (int k = 0; k < 4; k++) { u64 i = base + threadIdx.x * 4 + k; f[k] = i < n ? in[i] : 0; s += f[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (u32 off = 1; off < 256; off <<= 1) {
    u32 v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = sums[blockIdx.x] + sh[threadIdx.x] - s;
  for (int k = 0; k < 4; k++) { u64 i = base + threadIdx.x * 4 + k; if (i < n) { out[i] = run; run += f[k]; } }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// round 0 bookkeeping: ioff[B] = w*n + base[B], icnt = tot; also the maximum bucket size (for the round count)
__global__ __launch_bounds__(256) void init_kernel(u32* __restrict__ ioff, const u32* __restrict__ base, const u32* __restrict__ tot,
                                                   msm::Jobs jobs, u32 nb, u32 W, u64 NB, u32* __restrict__ maxcnt) {
  u64 B = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 m = 0;
  if (B < NB) {
    u32 job = (u32)(B / ((u64)W * nb)), w = (u32)((B / nb) % W);
    ioff[B] = (u32)(jobs.ent_off[job] + (u64)w * jobs.n[job]) + base[B];
    m = tot[B];
  }
  // block max -> one atomic per wave
  for (int off = 32; off > 0; off >>= 1) { u32 o = __shfl_down(m, off); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(maxcnt, m);
}
__global__ __launch_bounds__(256) void next_counts_kernel(u32* __restrict__ ocnt, const u32* __restrict__ icnt, u64 NB) {
  u64 B = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (B < NB) ocnt[B] = (icnt[B] + 1) >> 1;
}

struct Round {
  msm::Jobs jobs;          // round 0: SRS points per job
  u32 wnb;                 // W * nb (buckets per job)
  const u32* sorted;       // round 0: entries (index | sign << 31)
  const G1Affine* pin;     // rounds >= 1: input points
  const u32* ioff;         // [NB] first input element of bucket B
  const u32* icnt;         // [NB] number of input elements
  const u32* ooff;         // [NB + 1] first output element of bucket B (exclusive scan of ceil(icnt / 2))
  u64 NB;
  u64 E;                   // number of outputs = ooff[NB]
  u64 T;                   // threads; thread t owns outputs t, t + T, ...
  int first;               // 1 = round 0
};

__device__ __forceinline__ bool is_inf(const G1Affine& p) { return p.x.is_zero() && p.y.is_zero(); }

__device__ __forceinline__ G1Affine load_in(const Round& r, u64 idx, u32 B) {
  if (r.first) {
    u32 e = r.sorted[idx];
    G1Affine p = g1_load_affine(r.jobs.bases[B / r.wnb] + (e & 0x7fffffffu));
    if (e & 0x80000000u) p.y = ff_neg(p.y);
    return p;
  }
  return g1_load_affine(r.pin + idx);
}

// kind: 0 = result is a (copy), 1 = result is b (copy), 2 = infinity, 3 = add (d = x2 - x1), 4 = double (d = 2 y1)
__device__ __forceinline__ int classify(const G1Affine& a, const G1Affine& b, bool has2, Fq& d) {
  if (!has2 || is_inf(b)) return 0;
  if (is_inf(a)) return 1;
  Fq dx = ff_sub(b.x, a.x);
  if (!dx.is_zero()) { d = dx; return 3; }
  if (a.y == b.y && !a.y.is_zero()) { d = ff_dbl(a.y); return 4; }
  return 2;
}

// owner bucket of output o: last B with ooff[B] <= o
__device__ __forceinline__ u32 find_bucket(const u32* __restrict__ ooff, u64 NB, u32 o) {
  u64 lo = 0, hi = NB;          // invariant: ooff[lo] <= o < ooff[hi]
  while (hi - lo > 1) {
    u64 mid = (lo + hi) >> 1;
    if (ooff[mid] <= o) lo = mid; else hi = mid;
  }
  return (u32)lo;
}

__global__ __launch_bounds__(TPB) void fwd_kernel(Round r, u32* __restrict__ ob, Fq* __restrict__ pre, Fq* __restrict__ prod) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= r.T) return;
  Fq run = Fq::one();
  for (u64 o = t; o < r.E; o += r.T) {
    u32 B = find_bucket(r.ooff, r.NB, (u32)o);
    ob[o] = B;
    u32 j = (u32)o - r.ooff[B];
    u64 i0 = (u64)r.ioff[B] + 2ull * j;
    bool has2 = 2 * j + 1 < r.icnt[B];
    G1Affine a = load_in(r, i0, B);
    G1Affine b = a;
    if (has2) b = load_in(r, i0 + 1, B);
    Fq d;
    int kind = classify(a, b, has2, d);
    ff_store(pre + o, run);
    if (kind >= 3) run = ff_mul(run, d);
  }
  ff_store(prod + t, run);
}

// prod[i] <- 1 / prod[i]   (no zeros can occur: every factor is a non-zero denominator)
constexpr int INV_CH = 32;
__global__ __launch_bounds__(64) void inv_kernel(Fq* __restrict__ prod, Fq* __restrict__ scratch, u64 T) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo = t * INV_CH;
  if (lo >= T) return;
  u64 hi = lo + INV_CH; if (hi > T) hi = T;
  Fq acc = Fq::one();
  for (u64 i = lo; i < hi; i++) { ff_store(scratch + i, acc); acc = ff_mul(acc, ff_load(prod + i)); }
  Fq inv = ff_inv(acc);
  for (u64 i = hi; i-- > lo;) {
    Fq v = ff_load(prod + i);
    ff_store(prod + i, ff_mul(inv, ff_load(scratch + i)));
    inv = ff_mul(inv, v);
  }
}

__global__ __launch_bounds__(TPB) void bwd_kernel(Round r, const u32* __restrict__ ob, const Fq* __restrict__ pre,
                                                  const Fq* __restrict__ prodinv, G1Affine* __restrict__ pout) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= r.T || t >= r.E) return;
  Fq inv = ff_load(prodinv + t);
  // last output of this thread's chain
  u64 last = t + ((r.E - 1 - t) / r.T) * r.T;
  for (u64 o = last;; o -= r.T) {
    u32 B = ob[o];
    u32 j = (u32)o - r.ooff[B];
    u64 i0 = (u64)r.ioff[B] + 2ull * j;
    bool has2 = 2 * j + 1 < r.icnt[B];
    G1Affine a = load_in(r, i0, B);
    G1Affine b = a;
    if (has2) b = load_in(r, i0 + 1, B);
    Fq d;
    int kind = classify(a, b, has2, d);
    G1Affine res;
    if (kind == 0) res = a;
    else if (kind == 1) res = b;
    else if (kind == 2) { res.x = Fq::zero(); res.y = Fq::zero(); }
    else {
      Fq dinv = ff_mul(inv, ff_load(pre + o));
      inv = ff_mul(inv, d);
      Fq num;
      if (kind == 3) num = ff_sub(b.y, a.y);
      else { Fq xx = ff_sqr(a.x); num = ff_add(ff_dbl(xx), xx); }
      Fq lam = ff_mul(num, dinv);
      Fq x3 = ff_sub(ff_sub(ff_sqr(lam), a.x), b.x);
      res.x = x3;
      res.y = ff_sub(ff_mul(lam, ff_sub(a.x, x3)), a.y);
    }
    ff_store(&pout[o].x, res.x);
    ff_store(&pout[o].y, res.y);
    if (o < r.T) break;
  }
}

// final: bucket B's sum is pin[ooff[B]] if it has a point, else the identity -> XYZZ bucket array for reduce
__global__ __launch_bounds__(256) void to_buckets_kernel(G1Xyzz* __restrict__ buckets, const G1Affine* __restrict__ pin,
                                                         const u32* __restrict__ off, const u32* __restrict__ cnt, u64 NB) {
  u64 B = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (B >= NB) return;
  G1Xyzz r = G1Xyzz::identity();
  if (cnt[B]) {
    G1Affine p = g1_load_affine(pin + off[B]);
    if (!is_inf(p)) { r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one(); }
  }
  g1_store_xyzz(buckets + B, r);
}
// same when no round ran at all (every bucket has <= 1 entry): read straight from the sorted lists
__global__ __launch_bounds__(256) void to_buckets0_kernel(G1Xyzz* __restrict__ buckets, msm::Jobs jobs, u32 wnb,
                                                          const u32* __restrict__ sorted, const u32* __restrict__ ioff,
                                                          const u32* __restrict__ cnt, u64 NB) {
  u64 B = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (B >= NB) return;
  G1Xyzz r = G1Xyzz::identity();
  if (cnt[B]) {
    u32 e = sorted[ioff[B]];
    G1Affine p = g1_load_affine(jobs.bases[B / wnb] + (e & 0x7fffffffu));
    if (e & 0x80000000u) p.y = ff_neg(p.y);
    r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one();
  }
  g1_store_xyzz(buckets + B, r);
}

}  // namespace msmtree
