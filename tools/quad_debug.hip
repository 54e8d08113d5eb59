// Debug harness of the one-point-per-quad group law (msm_fb_quad.cuh): which sub-test disagrees with the one-lane form.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I marlin_amd/csrc tools/quad_debug.hip -o tools/quad_debug
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ff.cuh"
#include "g1.cuh"
#include "msm.cuh"
#include "msm_fb.cuh"
using namespace msmfb;

__global__ void perm_kernel(u32* out) {
  const u32 t = threadIdx.x;
  out[t] = qperm<MH_QP(0, 0, 1, 1)>(t);
  out[64 + t] = qperm<MH_QP(2, 2, 3, 3)>(t);
  out[128 + t] = qperm<MH_QP(0, 2, 2, 3)>(t);
  Fq30 a;
  for (int i = 0; i < Fq30::NL; i++) a.v[i] = t * 100 + i;
  Fq30 b = fperm<MH_QP(1, 1, 3, 3)>(a);
  out[192 + t] = b.v[5];
}

// fail[k] counts threads whose coordinate differs in sub-test k; dump = limbs of the first quad for sub-test 0
__global__ __launch_bounds__(256) void dbg_kernel(const Fq* __restrict__ in, u64 n, u32* __restrict__ fail, u32* __restrict__ dump) {
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
  auto diff = [&](const Fq30& x, const Fq30& y) { bool d = false; for (int k = 0; k < Fq30::NL; k++) d = d || x.v[k] != y.v[k]; return d; };
  const Fq30 pc = q30_pick(p30, role), qc = q30_pick(q30, role);
  // 0: pick / gather round trip
  { X30 g = q30_gather(pc); if (diff(q30_pick(g, role), pc) || diff(g.x, p30.x) || diff(g.y, p30.y) || diff(g.zz, p30.zz) || diff(g.zzz, p30.zzz)) atomicAdd(fail + 0, 1u); }
  // 1: generic add
  { X30 r = p30; x30_add(r, q30); Fq30 rq = pc; q30_add(rq, qc, role);
    if (diff(rq, q30_pick(r, role))) atomicAdd(fail + 1 + 8 * (1 + role), 1u), atomicAdd(fail + 1, 1u);
    if (i == 7) for (int k = 0; k < Fq30::NL; k++) { dump[role * 32 + k] = rq.v[k]; dump[role * 32 + 16 + k] = q30_pick(r, role).v[k]; } }
  // 2: doubling
  { X30 r = p30; x30_dbl(r); Fq30 rq = pc; q30_dbl(rq, role); if (diff(rq, q30_pick(r, role))) atomicAdd(fail + 2 + 8 * (1 + role), 1u), atomicAdd(fail + 2, 1u); }
  // 3: equal x
  { X30 r = p30; x30_add(r, p30); Fq30 rq = pc; q30_add(rq, pc, role); if (diff(rq, q30_pick(r, role))) atomicAdd(fail + 3, 1u); }
  // 4, 5, 6: identities
  { Fq30 rq = f30_zero(); q30_add(rq, pc, role); if (diff(rq, pc)) atomicAdd(fail + 4, 1u); }
  { Fq30 rq = pc; q30_add(rq, f30_zero(), role); if (diff(rq, pc)) atomicAdd(fail + 5, 1u); }
  { Fq30 rq = f30_zero(); q30_dbl(rq, role); if (diff(rq, f30_zero())) atomicAdd(fail + 6, 1u); }
  // 7: inline-free level-1 product only: lane r's product against the one-lane computation
  {
    const bool odd = role & 1;
    const Fq30 M1 = f30_mul(fsel(odd, fperm<MH_QP(0, 0, 1, 1)>(qc), fperm<MH_QP(0, 0, 1, 1)>(pc)),
                            fsel(odd, fperm<MH_QP(2, 2, 3, 3)>(pc), fperm<MH_QP(2, 2, 3, 3)>(qc)));
    const Fq30 want = role == 0 ? f30_mul(p30.x, q30.zz) : role == 1 ? f30_mul(q30.x, p30.zz) : role == 2 ? f30_mul(p30.y, q30.zzz) : f30_mul(q30.y, p30.zzz);
    if (diff(M1, want)) atomicAdd(fail + 7, 1u);
  }
}

int main() {
  u32* d_out; hipMalloc(&d_out, 256 * 4);
  perm_kernel<<<1, 64>>>(d_out);
  u32 h[256]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; r++) { printf("perm %d:", r); for (int i = 0; i < 8; i++) printf(" %u", h[r * 64 + i]); printf("\n"); }
  const u64 n = 1 << 12;
  std::vector<u32> hin((n + 1) * Fq::N);
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (size_t i = 0; i < hin.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hin[i] = (u32)(x >> 16); if (i % Fq::N == Fq::N - 1) hin[i] &= 0x0fffffffu; }
  Fq* d_in; hipMalloc(&d_in, hin.size() * 4); hipMemcpy(d_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  u32 *d_fail, *d_dump; hipMalloc(&d_fail, 64 * 4); hipMalloc(&d_dump, 128 * 4); hipMemset(d_fail, 0, 64 * 4); hipMemset(d_dump, 0, 128 * 4);
  dbg_kernel<<<(4 * n + 255) / 256, 256>>>(d_in, n, d_fail, d_dump);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  u32 f[64], dmp[128]; hipMemcpy(f, d_fail, sizeof(f), hipMemcpyDeviceToHost); hipMemcpy(dmp, d_dump, sizeof(dmp), hipMemcpyDeviceToHost);
  const char* names[8] = {"gather/pick", "add", "dbl", "equal-x", "O+p", "p+O", "2O", "level-1 product"};
  for (int k = 0; k < 8; k++) printf("%-16s mismatching lanes: %u of %llu\n", names[k], f[k], (unsigned long long)(4 * n));
  for (int r = 0; r < 4; r++) printf("add role %d: %u   dbl role %d: %u\n", r, f[1 + 8 * (1 + r)], r, f[2 + 8 * (1 + r)]);
  for (int r = 0; r < 4; r++) {
    printf("quad 7 role %d got :", r); for (int k = 0; k < 13; k++) printf(" %08x", dmp[r * 32 + k]); printf("\n");
    printf("quad 7 role %d want:", r); for (int k = 0; k < 13; k++) printf(" %08x", dmp[r * 32 + 16 + k]); printf("\n");
  }
  return 0;
}
