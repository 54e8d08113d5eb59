// Device-resident polynomial "glue" kernels over BLS12-381 Fr for the AHP prover rounds:
// everything /root/reference src/ahp/prover.rs does between its FFTs and commitments, so that
// a whole Marlin::prove runs without its polynomials leaving HBM.
// Each kernel cites the reference lines it stands in for.  All are HBM-streaming, one or two
// field multiplications per element.
#pragma once
#include "ff.cuh"

namespace poly {

constexpr int TPB = 256;
inline unsigned grid_for(u64 n, int tpb = TPB) { u64 g = (n + tpb - 1) / tpb; return (unsigned)(g ? g : 1); }

struct FrArg { Fr v; };   // pass-by-value scalar

// out[i] = sum_j coef[j] * src[j][i]  (src[j] has len[j] elements, treated as 0 beyond), i < n.
// Stands in for the coefficient-wise loops at prover.rs:473-479 (summed_z_m), 625-639 (a_poly),
// the LC polynomial construction and `p += (challenge, poly)` accumulations in
// ark-poly-commit's open_combinations / open (lib.rs:292), `mask + rhs` (prover.rs:546).
constexpr int MAX_TERMS = 8;
struct LinComb {
  const Fr* src[MAX_TERMS];
  u64 len[MAX_TERMS];
  Fr coef[MAX_TERMS];
  int is_one[MAX_TERMS];   // coefficient == 1: skip the multiplication
  int nterms;
};
__global__ __launch_bounds__(TPB) void lincomb_kernel(Fr* __restrict__ out, u64 n, LinComb lc) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr acc = Fr::zero();
  for (int j = 0; j < lc.nterms; j++) {
    if (i < lc.len[j]) {
      Fr v = ff_load(lc.src[j] + i);
      if (!lc.is_one[j]) v = ff_mul(v, lc.coef[j]);
      acc = ff_add(acc, v);
    }
  }
  ff_store(out + i, acc);
}

// out[i] = a[i] * b[i]                       (DensePolynomial `*` pointwise step, prover.rs:467,685)
__global__ __launch_bounds__(TPB) void mul_kernel(Fr* __restrict__ out, const Fr* __restrict__ a,
                                                  const Fr* __restrict__ b, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ff_store(out + i, ff_mul(ff_load(a + i), ff_load(b + i)));
}

// out[i] = a[i]*b[i] - c[i]*d[i]             (prover.rs:537-544)
__global__ __launch_bounds__(TPB) void mul_sub_mul_kernel(Fr* __restrict__ out, const Fr* __restrict__ a,
                                                          const Fr* __restrict__ b, const Fr* __restrict__ c,
                                                          const Fr* __restrict__ d, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr x = ff_mul(ff_load(a + i), ff_load(b + i));
  Fr y = ff_mul(ff_load(c + i), ff_load(d + i));
  ff_store(out + i, ff_sub(x, y));
}

// b[i] <- ec * a[i] * b[i] + ea * a[i] + eb * b[i]: the evaluations of summed_z_m = eta_a z_a + eta_b z_b + eta_c z_a z_b
// (prover.rs:467-471) on the 4H domain from those of z_a and z_b
__global__ __launch_bounds__(TPB) void summed_evals_kernel(Fr* __restrict__ b, const Fr* __restrict__ a, FrArg ea, FrArg eb,
                                                           FrArg ec, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr va = ff_load(a + i), vb = ff_load(b + i);
  Fr r = ff_mul(ff_mul(va, vb), ec.v);
  r = ff_add(r, ff_add(ff_mul(va, ea.v), ff_mul(vb, eb.v)));
  ff_store(b + i, r);
}

// out[i] = in[i] * g^(+-i) with g^i = hi[i >> 11] * lo[i & 2047]: moves a coefficient vector to / from the coset g K
// (evaluations of p on g K = transform of the coefficients p_i g^i)
constexpr int COSET_LO_BITS = 11;
__global__ __launch_bounds__(TPB) void twist_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, const Fr* __restrict__ hi,
                                                    const Fr* __restrict__ lo, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr gi = ff_mul(ff_load(hi + (i >> COSET_LO_BITS)), ff_load(lo + (i & ((1u << COSET_LO_BITS) - 1))));
  ff_store(out + i, ff_mul(ff_load(in + i), gi));
}

// evaluations of h_2 on the coset g K (prover.rs:640-690 restated on a coset, see prover.hip):
// out = ((ea va + eb vb + ec vc) - (alpha beta - alpha row - beta col + row_col) * f) * scale
__global__ __launch_bounds__(TPB) void h2_coset_kernel(Fr* __restrict__ out, const Fr* __restrict__ f, const Fr* __restrict__ va,
                                                       const Fr* __restrict__ vb, const Fr* __restrict__ vc, const Fr* __restrict__ row,
                                                       const Fr* __restrict__ col, const Fr* __restrict__ row_col, FrArg ea, FrArg eb,
                                                       FrArg ec, FrArg alpha, FrArg beta, FrArg alpha_beta, FrArg scale, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr a = ff_add(ff_add(ff_mul(ea.v, ff_load(va + i)), ff_mul(eb.v, ff_load(vb + i))), ff_mul(ec.v, ff_load(vc + i)));
  Fr b = ff_add(ff_sub(ff_sub(alpha_beta.v, ff_mul(alpha.v, ff_load(row + i))), ff_mul(beta.v, ff_load(col + i))), ff_load(row_col + i));
  ff_store(out + i, ff_mul(ff_sub(a, ff_mul(b, ff_load(f + i))), scale.v));
}

// dst[off + i] += v[i]: folds a second coefficient vector, whose bases are the same SRS shifted by `off`, into one MSM
__global__ __launch_bounds__(TPB) void add_shifted_kernel(Fr* __restrict__ dst, const Fr* __restrict__ v, u64 off, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ff_store(dst + off + i, ff_add(ff_load(dst + off + i), ff_load(v + i)));
}
// dst[off + i] += s * v[i]
__global__ __launch_bounds__(TPB) void axpy_shifted_kernel(Fr* __restrict__ dst, const Fr* __restrict__ v, FrArg s, u64 off, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ff_store(dst + off + i, ff_add(ff_load(dst + off + i), ff_mul(s.v, ff_load(v + i))));
}
// p[0] += v
__global__ void add_const_kernel(Fr* __restrict__ p, FrArg v) {
  if (blockIdx.x == 0 && threadIdx.x == 0) ff_store(p, ff_add(ff_load(p), v.v));
}

// out[i] = s * a[i]
__global__ __launch_bounds__(TPB) void scale_kernel(Fr* __restrict__ out, const Fr* __restrict__ a, FrArg s, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ff_store(out + i, ff_mul(ff_load(a + i), s.v));
}

// z = x || w on device; w_ext zero padding is implicit in w_evals_kernel.
// w_poly_evals (prover.rs:340-348): k % ratio == 0 -> 0 else w_ext[k - k/ratio - 1] - x_evals[k]
__global__ __launch_bounds__(TPB) void w_evals_kernel(Fr* __restrict__ out, const Fr* __restrict__ witness, u64 n_wit,
                                                      const Fr* __restrict__ x_evals, u64 H, u64 ratio) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= H) return;
  Fr r = Fr::zero();
  if (k % ratio != 0) {
    u64 j = k - (k / ratio) - 1;
    Fr w = j < n_wit ? ff_load(witness + j) : Fr::zero();
    r = ff_sub(w, ff_load(x_evals + k));
  }
  ff_store(out + k, r);
}

// CSR sparse matrix-vector product z_M = M z   (prover.rs:256-276).  val == nullptr: all ones.
__global__ __launch_bounds__(TPB) void spmv_kernel(Fr* __restrict__ out, const u64* __restrict__ row_ptr,
                                                   const u32* __restrict__ col, const Fr* __restrict__ val,
                                                   const Fr* __restrict__ z, u64 nrows) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  Fr acc = Fr::zero();
  for (u64 e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
    Fr v = ff_load(z + col[e]);
    if (val) v = ff_mul(v, ff_load(val + e));
    acc = ff_add(acc, v);
  }
  ff_store(out + r, acc);
}

// arithmetize_matrix (constraint_systems.rs:125-262), one thread per entry k < |K| of the joint matrix (indexer.rs:83-102):
// entry k sits at (row r, column i) of the union of A, B, C; with the domain elements h_r = omega^r and
// c = omega^reindex(i) (the columns are reindexed by the input subdomain, constraint_systems.rs:150-163; the matrices are
// transposed, so the "row" polynomial interpolates the column element),
//     row(k) = c,  col(k) = h_r,  row_col(k) = c h_r,  val_M(k) = M[r][i] c / |H|
// (u_H(c, c)^-1 = c / |H| because c^|H| = 1).  Repeated entries of a row are summed; k >= nnz pads with omega^0 and 0.
// val == nullptr: all coefficients are one.
struct ArithMat { const u64* row_ptr; const u32* col; const Fr* val; };
__device__ __forceinline__ u64 reindex_by_subdomain_dev(u64 H, u64 X, u64 index) {
  const u64 period = H / X;
  if (index < X) return index * period;
  const u64 i = index - X, x = period - 1;
  return i + (i / x) + 1;
}
__global__ __launch_bounds__(TPB) void arithmetize_kernel(Fr* __restrict__ row_v, Fr* __restrict__ col_v, Fr* __restrict__ rc_v,
                                                          Fr* __restrict__ va, Fr* __restrict__ vb, Fr* __restrict__ vc,
                                                          const u32* __restrict__ jrow, const u32* __restrict__ jcol, u64 nnz, u64 K,
                                                          ArithMat A, ArithMat B, ArithMat C, FrArg omega, FrArg h_inv, u64 H, u64 X) {
  const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  Fr* vals[3] = {va, vb, vc};
  if (k >= nnz) {
    ff_store(row_v + k, Fr::one()); ff_store(col_v + k, Fr::one()); ff_store(rc_v + k, Fr::one());
    for (int q = 0; q < 3; q++) ff_store(vals[q] + k, Fr::zero());
    return;
  }
  const u32 r = jrow[k], i = jcol[k];
  const Fr c = ff_pow(omega.v, reindex_by_subdomain_dev(H, X, i));
  const Fr hr = ff_pow(omega.v, (u64)r);
  ff_store(row_v + k, c);
  ff_store(col_v + k, hr);
  ff_store(rc_v + k, ff_mul(c, hr));
  const Fr scale = ff_mul(c, h_inv.v);
  const ArithMat mats[3] = {A, B, C};
  for (int q = 0; q < 3; q++) {
    Fr acc = Fr::zero();
    for (u64 e = mats[q].row_ptr[r]; e < mats[q].row_ptr[r + 1]; e++)
      if (mats[q].col[e] == i) acc = ff_add(acc, mats[q].val ? ff_load(mats[q].val + e) : Fr::one());
    ff_store(vals[q] + k, ff_mul(acc, scale));
  }
}

// p[0] -= r, p[n] = r: adds r * (X^n - 1) to a polynomial with n coefficients (the hiding term r * v_H of
// prover.rs:352,360,366) without a host round trip
__global__ void blind_vanishing_kernel(Fr* __restrict__ p, u64 n, FrArg r) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ff_store(p, ff_sub(ff_load(p), r.v));
    ff_store(p + n, r.v);
  }
}

// out[i] = sum_k coef[k] omega^(i k), i < n = 2^log_n, for a polynomial with few coefficients (x_poly of a small public
// input, prover.rs:326): Horner at omega^i instead of a size-n transform
__global__ __launch_bounds__(TPB) void eval_small_poly_kernel(Fr* __restrict__ out, const Fr* __restrict__ coef, u32 ncoef,
                                                              const Fr* __restrict__ tw, u32 log_n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 n = 1ull << log_n;
  if (i >= n) return;
  Fr w;
  if (log_n == 0) w = Fr::one();
  else {
    u64 half = n >> 1;
    w = ff_load(tw + half + (i & (half - 1)));
    if (i >= half) w = ff_neg(w);
  }
  Fr acc = ff_load(coef + ncoef - 1);
  for (u32 k = ncoef - 1; k-- > 0;) acc = ff_add(ff_mul(acc, w), ff_load(coef + k));
  ff_store(out + i, acc);
}

// out[i] = x - omega^i, i < n (n = 2^log_n), omega^i read from the NTT twiddle table (level log_n
// holds omega^e for e < n/2; omega^(e + n/2) = -omega^e).     (mod.rs:313: elements().map(|y| x - y))
__global__ __launch_bounds__(TPB) void x_minus_elements_kernel(Fr* __restrict__ out, const Fr* __restrict__ tw,
                                                               FrArg x, u32 log_n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 n = 1ull << log_n;
  if (i >= n) return;
  Fr w;
  if (log_n == 0) w = Fr::one();
  else {
    u64 half = n >> 1;
    w = ff_load(tw + half + (i & (half - 1)));
    if (i >= half) w = ff_neg(w);
  }
  ff_store(out + i, ff_sub(x.v, w));
}

// Batch inversion (ark_ff::batch_inversion, call sites prover.rs:663, mod.rs:314), Montgomery's
// trick per thread over a contiguous chunk of CH elements, zeros left untouched; optionally
// multiplies every inverse by `scale` (mod.rs:316).  scratch: n elements.
constexpr int INV_CH = 32;
__global__ __launch_bounds__(TPB) void batch_inverse_kernel(Fr* __restrict__ data, Fr* __restrict__ scratch, u64 n,
                                                            FrArg scale, int do_scale) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo = t * INV_CH;
  if (lo >= n) return;
  u64 hi = lo + INV_CH; if (hi > n) hi = n;
  Fr acc = Fr::one();
  for (u64 i = lo; i < hi; i++) {
    Fr v = ff_load(data + i);
    ff_store(scratch + i, acc);            // product of the non-zero elements before i
    if (!v.is_zero()) acc = ff_mul(acc, v);
  }
  Fr inv = ff_inv(acc);
  if (do_scale) inv = ff_mul(inv, scale.v);
  for (u64 i = hi; i-- > lo;) {
    Fr v = ff_load(data + i);
    if (v.is_zero()) continue;
    Fr pre = ff_load(scratch + i);
    ff_store(data + i, ff_mul(inv, pre));
    inv = ff_mul(inv, v);
  }
}

// Two-level variant for large n: thread t of a block walks the elements base + t + TPB * j (coalesced), the per-thread
// totals are inverted by batch_inverse_kernel (a factor INV_CH fewer Fermat inversions), and a backward sweep applies them.
// scratch: n elements; totals: one per thread.
__global__ __launch_bounds__(TPB) void binv_fwd_kernel(const Fr* __restrict__ data, Fr* __restrict__ scratch, Fr* __restrict__ totals, u64 n) {
  const u64 base = (u64)blockIdx.x * (TPB * INV_CH) + threadIdx.x;
  Fr acc = Fr::one();
  for (int j = 0; j < INV_CH; j++) {
    const u64 i = base + (u64)TPB * j;
    if (i >= n) break;
    Fr v = ff_load(data + i);
    ff_store(scratch + i, acc);
    if (!v.is_zero()) acc = ff_mul(acc, v);
  }
  ff_store(totals + (u64)blockIdx.x * TPB + threadIdx.x, acc);
}
__global__ __launch_bounds__(TPB) void binv_bwd_kernel(Fr* __restrict__ data, const Fr* __restrict__ scratch,
                                                       const Fr* __restrict__ inv_totals, u64 n, FrArg scale, int do_scale) {
  const u64 base = (u64)blockIdx.x * (TPB * INV_CH) + threadIdx.x;
  if (base >= n) return;
  Fr inv = ff_load(inv_totals + (u64)blockIdx.x * TPB + threadIdx.x);
  if (do_scale) inv = ff_mul(inv, scale.v);
  int last = (int)((n - 1 - base) / TPB);
  if (last > INV_CH - 1) last = INV_CH - 1;
  for (int j = last; j >= 0; j--) {
    const u64 i = base + (u64)TPB * j;
    Fr v = ff_load(data + i);
    if (v.is_zero()) continue;
    ff_store(data + i, ff_mul(inv, ff_load(scratch + i)));
    inv = ff_mul(inv, v);
  }
}

// b evals on K (prover.rs:650-654): alpha*beta - alpha*row - beta*col + row_col
__global__ __launch_bounds__(TPB) void b_evals_kernel(Fr* __restrict__ out, const Fr* __restrict__ row,
                                                      const Fr* __restrict__ col, const Fr* __restrict__ row_col,
                                                      FrArg alpha, FrArg beta, FrArg alpha_beta, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr r = ff_mul(alpha.v, ff_load(row + i));
  Fr c = ff_mul(beta.v, ff_load(col + i));
  Fr v = ff_add(ff_sub(ff_sub(alpha_beta.v, r), c), ff_load(row_col + i));
  ff_store(out + i, v);
}

// denominators (prover.rs:660-662): (beta - row)(alpha - col)
__global__ __launch_bounds__(TPB) void denom_kernel(Fr* __restrict__ out, const Fr* __restrict__ row,
                                                    const Fr* __restrict__ col, FrArg alpha, FrArg beta, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ff_store(out + i, ff_mul(ff_sub(beta.v, ff_load(row + i)), ff_sub(alpha.v, ff_load(col + i))));
}

// f evals on K (prover.rs:670-677): inv * (ea*val_a + eb*val_b + ec*val_c)
__global__ __launch_bounds__(TPB) void f_evals_kernel(Fr* __restrict__ out, const Fr* __restrict__ inv,
                                                      const Fr* __restrict__ va, const Fr* __restrict__ vb,
                                                      const Fr* __restrict__ vc, FrArg ea, FrArg eb, FrArg ec, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = ff_add(ff_add(ff_mul(ea.v, ff_load(va + i)), ff_mul(eb.v, ff_load(vb + i))), ff_mul(ec.v, ff_load(vc + i)));
  ff_store(out + i, ff_mul(ff_load(inv + i), s));
}

// Division by the vanishing polynomial X^n - 1 (DensePolynomial::divide_by_vanishing_poly,
// prover.rs:353,550,686): q_i = sum_{j>=1} p_{i+jn}, r_i = p_i + q_i (i < n).  The recurrence
// q_i = p_{i+n} + q_{i+n} is a suffix sum along each residue class mod n.  Three phases over
// (column = i mod n, chunk of DIV_ROWS rows): chunk sums, per-column suffix scan of chunk sums,
// chunk-local suffix sums with carry.  p has len coefficients; q gets len - n (len > n).
constexpr int DIV_ROWS = 64;
// phase 1: partial[chunk * n + col] = sum of p[(row)*n + col] for rows in chunk (rows >= 1)
__global__ __launch_bounds__(TPB) void divvan_partial_kernel(Fr* __restrict__ partial, const Fr* __restrict__ p, u64 len,
                                                             u64 n, u64 nchunks) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nchunks) return;
  u64 col = t % n, chunk = t / n;
  u64 r0 = 1 + chunk * DIV_ROWS, r1 = r0 + DIV_ROWS;
  Fr acc = Fr::zero();
  for (u64 r = r0; r < r1; r++) {
    u64 idx = r * n + col;
    if (idx < len) acc = ff_add(acc, ff_load(p + idx));
  }
  ff_store(partial + t, acc);
}
// phase 2: per column, partial[chunk] <- sum of partials of LATER chunks (exclusive suffix).
// One block per column: each thread scans a contiguous run of chunks, the 256 run totals are scanned in LDS.
__global__ __launch_bounds__(TPB) void divvan_scan_kernel(Fr* __restrict__ partial, u64 n, u64 nchunks) {
  __shared__ uint4 sh[TPB * 2];
  Fr* s = reinterpret_cast<Fr*>(sh);
  const u64 col = blockIdx.x;
  const u64 per = (nchunks + TPB - 1) / TPB;
  // thread t owns chunks [lo, hi) counted from the TOP: reversed index rc = nchunks - 1 - c
  const u64 lo = threadIdx.x * per;
  u64 hi = lo + per; if (hi > nchunks) hi = nchunks;
  Fr tot = Fr::zero();
  for (u64 rc = lo; rc < hi; rc++) tot = ff_add(tot, ff_load(partial + (nchunks - 1 - rc) * n + col));
  s[threadIdx.x] = tot;
  __syncthreads();
  for (int off = 1; off < TPB; off <<= 1) {      // inclusive scan of run totals
    Fr v = (int)threadIdx.x >= off ? s[threadIdx.x - off] : Fr::zero();
    __syncthreads();
    s[threadIdx.x] = ff_add(s[threadIdx.x], v);
    __syncthreads();
  }
  Fr run = ff_sub(s[threadIdx.x], tot);            // sum of the runs above this one
  for (u64 rc = lo; rc < hi; rc++) {
    u64 idx = (nchunks - 1 - rc) * n + col;
    Fr v = ff_load(partial + idx);
    ff_store(partial + idx, run);
    run = ff_add(run, v);
  }
}
// same, one thread per column (many columns, few chunks)
__global__ __launch_bounds__(TPB) void divvan_scan_simple_kernel(Fr* __restrict__ partial, u64 n, u64 nchunks) {
  u64 col = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= n) return;
  Fr run = Fr::zero();
  for (u64 c = nchunks; c-- > 0;) {
    Fr v = ff_load(partial + c * n + col);
    ff_store(partial + c * n + col, run);
    run = ff_add(run, v);
  }
}
// phase 3: q[(r-1)*n + col] = sum_{r' >= r} p[r'*n + col]
__global__ __launch_bounds__(TPB) void divvan_final_kernel(Fr* __restrict__ q, const Fr* __restrict__ partial,
                                                           const Fr* __restrict__ p, u64 len, u64 n, u64 nchunks) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nchunks) return;
  u64 col = t % n, chunk = t / n;
  u64 r0 = 1 + chunk * DIV_ROWS, r1 = r0 + DIV_ROWS;
  Fr run = ff_load(partial + t);
  for (u64 r = r1; r-- > r0;) {
    u64 idx = r * n + col;
    if (idx < len) {
      run = ff_add(run, ff_load(p + idx));
      ff_store(q + (idx - n), run);
    }
  }
}

// mul_by_vanishing_poly + x_poly (prover.rs:512-515): z = w * (X^n - 1) + x
//   z[i] = (i >= n ? w[i-n] : 0) - (i < wlen ? w[i] : 0) + (i < xlen ? x[i] : 0),   i < wlen + n
__global__ __launch_bounds__(TPB) void z_poly_kernel(Fr* __restrict__ z, const Fr* __restrict__ w, u64 wlen, u64 n,
                                                     const Fr* __restrict__ x, u64 xlen) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wlen + n) return;
  Fr v = Fr::zero();
  if (i >= n) v = ff_load(w + (i - n));
  if (i < wlen) v = ff_sub(v, ff_load(w + i));
  if (i < xlen) v = ff_add(v, ff_load(x + i));
  ff_store(z + i, v);
}

// Polynomial evaluation p(z) (DensePolynomial::evaluate, lib.rs:279 via mod.rs:242-266):
// each thread Horner-evaluates a chunk of EV_CH coefficients and scales by z^(chunk start);
// partial sums are reduced per block, then by one final block.
constexpr int EV_CH = 16;
__global__ __launch_bounds__(TPB) void eval_partial_kernel(Fr* __restrict__ partial, const Fr* __restrict__ p, u64 len,
                                                           FrArg z) {
  __shared__ uint4 sh[TPB * 2];
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo = t * EV_CH;
  Fr acc = Fr::zero();
  if (lo < len) {
    u64 hi = lo + EV_CH; if (hi > len) hi = len;
    for (u64 i = hi; i-- > lo;) acc = ff_add(ff_mul(acc, z.v), ff_load(p + i));
    acc = ff_mul(acc, ff_pow(z.v, lo));
  }
  // block reduction
  Fr* s = reinterpret_cast<Fr*>(sh);
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int off = TPB / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s[threadIdx.x] = ff_add(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) ff_store(partial + blockIdx.x, s[0]);
}
__global__ __launch_bounds__(TPB) void sum_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, u64 n) {
  __shared__ uint4 sh[TPB * 2];
  Fr acc = Fr::zero();
  for (u64 i = threadIdx.x; i < n; i += TPB) acc = ff_add(acc, ff_load(in + i));
  Fr* s = reinterpret_cast<Fr*>(sh);
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int off = TPB / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s[threadIdx.x] = ff_add(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) ff_store(out, s[0]);
}

// Division by (X - z) with the remainder dropped: KZG10::compute_witness_polynomial
// (ark-poly-commit kzg10, reached from lib.rs:292).  q_{i-1} = p_i + z q_i is a suffix scan with
// multiplier z.  Chunks of LIN_CH coefficients: (1) chunk Horner values V_c (carry-in 0),
// (2) carry into chunk c = V_{c+1} + z^LIN_CH * carry_{c+1}: the same recurrence over chunks,
// solved recursively (host drives levels), (3) chunk-local scan seeded with the carry.
constexpr int LIN_CH = 16;
// level kernel 1: V[c] = sum_{i in chunk c} p[i] * z^(i - lo_c)   (Horner over the chunk)
__global__ __launch_bounds__(TPB) void divlin_chunk_kernel(Fr* __restrict__ V, const Fr* __restrict__ p, u64 len,
                                                           FrArg z) {
  u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo = c * LIN_CH;
  if (lo >= len) return;
  u64 hi = lo + LIN_CH; if (hi > len) hi = len;
  Fr acc = Fr::zero();
  for (u64 i = hi; i-- > lo;) acc = ff_add(ff_mul(acc, z.v), ff_load(p + i));
  ff_store(V + c, acc);
}
// serial solve at the top level: carry[c] = value carried INTO chunk c from above
//   carry[last] = 0; carry[c] = V[c+1] + zc * carry[c+1]
__global__ void divlin_top_kernel(Fr* __restrict__ carry, const Fr* __restrict__ V, u64 nchunks, FrArg zc) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fr run = Fr::zero();
  for (u64 c = nchunks; c-- > 0;) {
    ff_store(carry + c, run);
    run = ff_add(ff_load(V + c), ff_mul(zc.v, run));
  }
}
// given carries INTO each group of LIN_CH chunks (gcarry), expand to per-chunk carries:
// within group g: carry[c] for c from top of group down
__global__ __launch_bounds__(TPB) void divlin_expand_kernel(Fr* __restrict__ carry, const Fr* __restrict__ V,
                                                            const Fr* __restrict__ gcarry, u64 nchunks, FrArg zc) {
  u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo = g * LIN_CH;
  if (lo >= nchunks) return;
  u64 hi = lo + LIN_CH; if (hi > nchunks) hi = nchunks;
  Fr run = ff_load(gcarry + g);
  for (u64 c = hi; c-- > lo;) {
    ff_store(carry + c, run);
    run = ff_add(ff_load(V + c), ff_mul(zc.v, run));
  }
}
// final: q[i-1] = p[i] + z*q[i] inside chunk c seeded with carry[c]; q has len-1 coefficients.
__global__ __launch_bounds__(TPB) void divlin_final_kernel(Fr* __restrict__ q, const Fr* __restrict__ p,
                                                           const Fr* __restrict__ carry, u64 len, FrArg z) {
  u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 lo = c * LIN_CH;
  if (lo >= len) return;
  u64 hi = lo + LIN_CH; if (hi > len) hi = len;
  Fr run = ff_load(carry + c);          // = q[hi-1]
  for (u64 i = hi; i-- > lo;) {
    // q[i-1] = p[i] + z*q[i], where q[i] for i = hi-1 is the carry
    run = ff_add(ff_load(p + i), ff_mul(z.v, run));
    if (i >= 1) ff_store(q + (i - 1), run);
  }
}

// calculate_t (prover.rs:411-428) as a two-phase segmented sum over entries sorted by output
// index.  items: (first entry, count, output k, matrix m); entries: (row, coeff or one).
struct TItem { u64 first; u32 count; u32 k; u32 m; u32 pad; };
__global__ __launch_bounds__(TPB) void t_items_kernel(Fr* __restrict__ partial, const TItem* __restrict__ items, u64 nitems,
                                                      const u32* __restrict__ erow, const Fr* __restrict__ ecoef,
                                                      const Fr* __restrict__ r_alpha, FrArg eta_a, FrArg eta_b, FrArg eta_c) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nitems) return;
  TItem it = items[t];
  Fr acc = Fr::zero();
  for (u32 e = 0; e < it.count; e++) {
    Fr v = ff_load(r_alpha + erow[it.first + e]);
    if (ecoef) v = ff_mul(v, ff_load(ecoef + it.first + e));
    acc = ff_add(acc, v);
  }
  Fr eta = it.m == 0 ? eta_a.v : (it.m == 1 ? eta_b.v : eta_c.v);
  ff_store(partial + t, ff_mul(acc, eta));
}
// out[k] = sum of partial[item_ptr[k] .. item_ptr[k+1])   (used at two levels so no thread sums more than ~128 values)
__global__ __launch_bounds__(TPB) void t_sum_kernel(Fr* __restrict__ out, const Fr* __restrict__ partial,
                                                    const u64* __restrict__ item_ptr, u64 H) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= H) return;
  Fr acc = Fr::zero();
  for (u64 i = item_ptr[k]; i < item_ptr[k + 1]; i++) acc = ff_add(acc, ff_load(partial + i));
  ff_store(out + k, acc);
}

}  // namespace poly
