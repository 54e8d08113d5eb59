// Bucket reduction with ONE POINT PER QUAD OF LANES (round 3).
//
// Why: the segment reduction of msm_fb.cuh (reduce1_30 / reduce2_30) is a chain of dependent XYZZ additions -- 2 seg
// running-sum additions plus a ~19-bit double-and-add per segment, then tree levels -- run at one wave per SIMD, and a
// lone wave needs ~17 us per addition (14 dependent field multiplications of ~420 VALU instructions each).  The chain
// length, not the amount of work, sets the time: 1.5 ms per launch on one GPU (4 launches per proof) and still 0.8 ms
// on a rank of 8 that owns an eighth of the buckets (profiles/r03o_sim_5_8_last_prove_kernels.txt: 5.7 of 20 ms).
// Here the four lanes of a DPP quad hold the four coordinates (X, Y, ZZ, ZZZ) of one point and the 14 multiplications
// of an addition run as FOUR levels of one multiplication per lane (13 of the 16 slots do useful work); operands move
// between the lanes of a quad with v_mov_b32 quad_perm (full-rate VALU, no LDS).  The same formulas in the same order
// as x30_add / x30_dbl (EFD add-2008-s, dbl-2008-s-1), so every coordinate is limb-for-limb what the one-lane form
// computes (mh_selftest_fq30 checks that on the device).
//
// Measured (profiles/r03p_*): 2,807 instructions per addition instead of 7,099, but 13 us instead of 17 us,
// not 7 -- a lone wave with one multiplication in flight cannot interleave anything with a dependent v_mad_u64_u32.  The
// tree stage (reduce2_q_kernel) of a bucket-range shard gains 0.5 ms per proof and is what FbRun::reduce uses there; the
// segment stage (reduce1_q_kernel) loses at every size and runs only under MH_FB_QUAD=2 (tests keep it honest).
//
//   lane role        0        1          2        3
//   level 1 (add)    U1       U2         S1       S2            P = U2 - U1 (lanes 0, 1), R = S2 - S1 (lanes 2, 3)
//   level 2          PP       ZZ1 ZZ2    RR       ZZZ1 ZZZ2
//   level 3          Q        ZZ3        PPP      PPP           X3 = RR - PPP - 2 Q (every lane)
//   level 4          --       R (Q - X3) S1 PPP   ZZZ3          Y3 = lane 1 - lane 2
#pragma once

namespace msmfb {

#define MH_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

// MH_QFPERM: 1 = the builtin as it is (hipcc folds some of the moves into v_subrev_u32_dpp); 2 = the builtin behind an
// empty asm (no folding, the compiler still pads the read-after-write hazard); 3 = the move itself as inline asm with its
// own two wait states in front (a VALU result must not be read through DPP earlier)
#ifndef MH_QFPERM
#define MH_QFPERM 2
#endif
template <int CTRL>
__device__ __forceinline__ u32 qperm(u32 v) {
#if MH_QFPERM == 3
  u32 r;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf"
               : "=&v"(r) : "v"(v), "n"(CTRL & 3), "n"((CTRL >> 2) & 3), "n"((CTRL >> 4) & 3), "n"((CTRL >> 6) & 3));
  return r;
#else
  u32 r = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
#if MH_QFPERM == 2
  asm volatile("" : "+v"(r));
#endif
  return r;
#endif
}
template <int CTRL>
__device__ __forceinline__ Fq30 fperm(const Fq30& a) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) r.v[i] = qperm<CTRL>(a.v[i]);
  return r;
}
// c ? a : b, limb by limb
__device__ __forceinline__ Fq30 fsel(bool c, const Fq30& a, const Fq30& b) {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
__device__ __forceinline__ Fq30 f30_zero() {
  Fq30 r;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ u32 f30_or(const Fq30& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < Fq30::NL; i++) o |= a.v[i];
  return o;
}
// the identity is ZZ = 0 exactly (lane 2 of the quad)
__device__ __forceinline__ bool q30_is_identity(const Fq30& a) { return qperm<MH_QP(2, 2, 2, 2)>(f30_or(a)) == 0; }

// a quad's point as a whole point in every lane / a whole point's coordinate of this lane
__device__ __forceinline__ X30 q30_gather(const Fq30& a) {
  X30 r;
  r.x = fperm<MH_QP(0, 0, 0, 0)>(a); r.y = fperm<MH_QP(1, 1, 1, 1)>(a);
  r.zz = fperm<MH_QP(2, 2, 2, 2)>(a); r.zzz = fperm<MH_QP(3, 3, 3, 3)>(a);
  return r;
}
__device__ __forceinline__ Fq30 q30_pick(const X30& p, u32 role) {
  return fsel(role == 0, p.x, fsel(role == 1, p.y, fsel(role == 2, p.zz, p.zzz)));
}
// the equal-x case (the same point, or opposite points): every lane of the quad treats the whole points the way the one-lane law
// does (x30_add_inl: the 30-bit doubling, or the identity) -- limb for limb the same result
__device__ __noinline__ void q30_add_slow(Fq30& a, const Fq30& b, u32 role) {
  X30 A = q30_gather(a), B = q30_gather(b);
  const Fq30 R = f30_sub<2>(f30_mul(B.y, A.zzz), f30_mul(A.y, B.zzz));
  if (f30_is_zero(R)) x30_dbl(A); else A = x30_identity();
  a = q30_pick(A, role);
}

// A += B for the quad's points (role = lane & 3 holds coordinate `role`); control flow is uniform over the quad
__device__ __noinline__ void q30_add(Fq30& A, const Fq30& B, u32 role) {
  if (q30_is_identity(B)) return;
  if (q30_is_identity(A)) { A = B; return; }
  const bool odd = role & 1;
  // level 1: [X1 ZZ2, X2 ZZ1, Y1 ZZZ2, Y2 ZZZ1]
  const Fq30 M1 = f30_mul(fsel(odd, fperm<MH_QP(0, 0, 1, 1)>(B), fperm<MH_QP(0, 0, 1, 1)>(A)),
                          fsel(odd, fperm<MH_QP(2, 2, 3, 3)>(A), fperm<MH_QP(2, 2, 3, 3)>(B)));
  // [P, P, R, R]
  const Fq30 D = f30_sub<2>(fperm<MH_QP(1, 1, 3, 3)>(M1), fperm<MH_QP(0, 0, 2, 2)>(M1));
  if (__builtin_expect(qperm<MH_QP(0, 0, 0, 0)>((u32)f30_is_zero(D)) != 0, 0)) { q30_add_slow(A, B, role); return; }
  // level 2: [P P, ZZ1 ZZ2, R R, ZZZ1 ZZZ2]
  const Fq30 M2 = f30_mul(fsel(odd, fperm<MH_QP(0, 2, 2, 3)>(A), D), fsel(odd, fperm<MH_QP(0, 2, 2, 3)>(B), D));
  // level 3: [U1 PP, (ZZ1 ZZ2) PP, P PP, P PP]
  const Fq30 M3 = f30_mul(fsel(role == 0, M1, fsel(role == 1, M2, fperm<MH_QP(0, 0, 0, 0)>(D))), fperm<MH_QP(0, 0, 0, 0)>(M2));
  const Fq30 q = fperm<MH_QP(0, 0, 0, 0)>(M3);
  const Fq30 X3 = f30_sub2<3>(f30_sub<2>(fperm<MH_QP(2, 2, 2, 2)>(M2), fperm<MH_QP(2, 2, 2, 2)>(M3)), q);
  // level 4: [--, R (Q - X3), S1 PPP, (ZZZ1 ZZZ2) PPP]
  const Fq30 M4 = f30_mul(fsel(role <= 1, fperm<MH_QP(2, 2, 2, 2)>(D), fsel(role == 2, M1, M2)),
                          fsel(role <= 1, f30_sub<8>(q, X3), M3));
  const Fq30 y3 = f30_sub<2>(M4, fperm<MH_QP(0, 2, 2, 3)>(M4));            // lane 1: R (Q - X3) - S1 PPP
  A = fsel(role == 0, X3, fsel(role == 1, y3, fsel(role == 2, fperm<MH_QP(0, 1, 1, 3)>(M3), M4)));
}

//   level 1 (dbl)    XX       V = U^2     --       --           U = 2 Y, M = 3 XX
//   level 2          S = X V  W = U V     V ZZ     M M          X3 = M M - 2 S (every lane)
//   level 3          M (S-X3) W Y         --       W ZZZ        Y3 = lane 0 - lane 1
__device__ __noinline__ void q30_dbl(Fq30& A, u32 role) {
  if (q30_is_identity(A)) return;
  const Fq30 U = fsel(role == 1, f30_dbl(A), A);                           // lane 1: 2 Y; the others: their coordinate
  const Fq30 M1 = f30_mul(U, U);
  const Fq30 xx = fperm<MH_QP(0, 0, 0, 0)>(M1);
  const Fq30 Mv = f30_add(f30_dbl(xx), xx);
  const Fq30 M2 = f30_mul(fsel(role == 3, Mv, U), fsel(role == 3, Mv, fperm<MH_QP(1, 1, 1, 1)>(M1)));
  const Fq30 s = fperm<MH_QP(0, 0, 0, 0)>(M2);
  const Fq30 X3 = f30_sub2<3>(fperm<MH_QP(3, 3, 3, 3)>(M2), s);
  const Fq30 M3 = f30_mul(fsel(role == 0, Mv, fsel(role == 1, M2, fperm<MH_QP(1, 1, 1, 1)>(M2))),
                          fsel(role == 0, f30_sub<8>(s, X3), A));
  const Fq30 y3 = f30_sub<2>(fperm<MH_QP(0, 0, 2, 3)>(M3), fperm<MH_QP(1, 1, 2, 3)>(M3));   // lane 1: M (S - X3) - W Y
  A = fsel(role == 0, X3, fsel(role == 1, y3, fsel(role == 2, M2, M3)));
}

// ---- reduce1, one segment per quad: the arguments and the result are those of reduce1_30_kernel ------------------
__global__ __launch_bounds__(256) void reduce1_q_kernel(const G1Xyzz30* __restrict__ buckets, G1Xyzz30* __restrict__ segsum, u32 nb,
                                                        u32 nseg, u32 njobs, u32 seg, u32 pbuckets, Own own,
                                                        const u32* __restrict__ largest, u32 skew_limit) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 gid = t >> 2, role = t & 3;
  if (gid >= njobs * nseg) return;
  if (*largest > skew_limit) return;      // skewed batch: see reduce1_30_kernel
  const u32 w = gid / nseg, s = gid % nseg;
  const u32 l = s * seg;                                          // index among the owned buckets
  const u32 lo = (own.first + (l / pbuckets) * own.stride) * pbuckets + l % pbuckets;
  const u32 hi = lo + seg;
  Fq30 running = f30_zero(), acc = f30_zero();
  const G1Xyzz30* B = buckets + (u64)w * nb;
  for (u32 b = hi; b-- > lo;) {
    const Fq30 tc = load30(B[b].c[role]);
    q30_add(running, tc, role);
    q30_add(acc, running, role);
  }
  if (lo) {
    Fq30 m = f30_zero();
    const int top = 31 - __clz(lo);
    for (int bit = top; bit >= 0; bit--) {
      q30_dbl(m, role);
      if ((lo >> bit) & 1) q30_add(m, running, role);
    }
    q30_add(acc, m, role);
  }
  store30(segsum[gid].c[role], acc);
}

// ---- reduce2, 64 quads per block: arguments and results of reduce2_30_kernel --------------------------------------
__global__ __launch_bounds__(256) void reduce2_q_kernel(const G1Xyzz30* __restrict__ in, G1Xyzz30* __restrict__ out30,
                                                        G1Xyzz* __restrict__ out_std, u32 nseg, int last) {
  extern __shared__ __attribute__((aligned(16))) u32 lds30q[];
  G1Xyzz30* sh = reinterpret_cast<G1Xyzz30*>(lds30q);
  const u32 q = threadIdx.x >> 2, role = threadIdx.x & 3;
  const u32 w = blockIdx.y, chunks = gridDim.x;
  const u32 per = (nseg + chunks - 1) / chunks;
  const u32 lo = blockIdx.x * per;
  u32 hi = lo + per; if (hi > nseg) hi = nseg;
  Fq30 acc = f30_zero();
  for (u32 s = lo + q; s < hi; s += 64) {
    const Fq30 tc = load30(in[(u64)w * nseg + s].c[role]);
    q30_add(acc, tc, role);
  }
  store30(sh[q].c[role], acc);
  __syncthreads();
  for (u32 off = 32; off > 0; off >>= 1) {
    if (q < off) {
      Fq30 a = load30(sh[q].c[role]);
      const Fq30 b = load30(sh[q + off].c[role]);
      q30_add(a, b, role);
      store30(sh[q].c[role], a);
    }
    __syncthreads();
  }
  if (q == 0) {
    const Fq30 r = load30(sh[0].c[role]);
    const u64 o = (u64)w * chunks + blockIdx.x;
    if (last) {
      const bool ident = q30_is_identity(r);
      const Fq v = ident ? Fq::zero() : f30_to_fq(r);
      ff_store(reinterpret_cast<Fq*>(out_std + o) + role, v);
    } else {
      store30(out30[o].c[role], r);
    }
  }
}

// ---- self-test: the quad forms against x30_add / x30_dbl on pseudo-points, coordinate by coordinate ----------------
__global__ __launch_bounds__(256) void selftest30_quad_kernel(const Fq* __restrict__ in, u64 n, u32* __restrict__ bad) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 i = t >> 2;
  const u32 role = (u32)t & 3;
  if (i >= n) return;
  Fq a = ff_mul(ff_load(in + i), Fq::one()), b = ff_mul(ff_load(in + i + 1), Fq::one());
  G1Xyzz p, q;
  p.x = a; p.y = b; p.zz = ff_sqr(b); p.zzz = ff_mul(p.zz, b);
  q.x = b; q.y = ff_add(a, b); q.zz = ff_sqr(a); q.zzz = ff_mul(q.zz, a);
  if (p.zz.is_zero() || q.zz.is_zero()) return;
  const X30 p30 = x30_from_std(p), q30 = x30_from_std(q);
  bool ok = true;
  auto same = [&](const Fq30& x, const Fq30& y) { for (int k = 0; k < Fq30::NL; k++) ok = ok && x.v[k] == y.v[k]; };
  const Fq30 pc = q30_pick(p30, role), qc = q30_pick(q30, role);
  { X30 r = p30; x30_add(r, q30); Fq30 rq = pc; q30_add(rq, qc, role); same(rq, q30_pick(r, role)); }           // generic
  { X30 r = p30; x30_dbl(r); Fq30 rq = pc; q30_dbl(rq, role); same(rq, q30_pick(r, role)); }                   // doubling
  { X30 r = p30; x30_add(r, p30); Fq30 rq = pc; q30_add(rq, pc, role); same(rq, q30_pick(r, role)); }          // equal x
  { Fq30 rq = f30_zero(); q30_add(rq, pc, role); same(rq, pc); }                                               // O + p
  { Fq30 rq = pc; q30_add(rq, f30_zero(), role); same(rq, pc); }                                               // p + O
  { Fq30 rq = f30_zero(); q30_dbl(rq, role); same(rq, f30_zero()); }                                           // 2 O
  // a chain as the reduction runs it: running / acc over three points, then a double-and-add
  {
    X30 run = x30_identity(), acc = x30_identity();
    Fq30 runq = f30_zero(), accq = f30_zero();
    X30 d30 = p30; x30_dbl(d30);
    const X30 pts[3] = {p30, q30, d30};
    for (int k = 0; k < 3; k++) {
      x30_add(run, pts[k]); x30_add(acc, run);
      q30_add(runq, q30_pick(pts[k], role), role); q30_add(accq, runq, role);
    }
    X30 m = x30_identity(); Fq30 mq = f30_zero();
    for (int bit = 4; bit >= 0; bit--) {
      x30_dbl(m); q30_dbl(mq, role);
      if ((0x15 >> bit) & 1) { x30_add(m, run); q30_add(mq, runq, role); }
    }
    x30_add(acc, m); q30_add(accq, mq, role);
    same(accq, q30_pick(acc, role));
  }
  if (!ok) atomicAdd(bad, 1u);
}

}  // namespace msmfb
