#!/usr/bin/env python3
"""Generate mont_mul_gen.inc: Montgomery multiplication for N x 32-bit limbs as
product-scanning (Comba) columns of `v_mad_u64_u32` + `v_addc_co_u32` inline-asm
statements for gfx950.

Why generated asm: hipcc turns every C++ formulation tried (CIOS with u64, Comba
with overflow builtins) into 4-5 VALU instructions per 32x32 limb product
(v_mov pairs to widen addends, v_cmp_lt_u64 + v_cndmask for the carry).  The
hardware sequence is two: a 64-bit multiply-add with carry-out to VCC and an
add-with-carry into the third accumulator word.  One asm statement per column
part keeps hipcc's post-asm `s_nop` padding to ~2 per column instead of one per
product.  The modulus words are "s" operands: compile-time constants that hipcc
materialises into SGPRs (VOP3 on gfx9 takes one SGPR source; 32-bit literals
are not encodable in VOP3).

Run: python3 gen_mont.py > mont_mul_gen.inc   (the output is committed)
"""
import sys


class Stmt:
    """one asm statement accumulating into (lo:64, hi:32) = operands %0, %1."""

    def __init__(self):
        self.body = []
        self.ins = []
        self.idx = {}

    def op(self, expr, cons):
        key = (expr, cons)
        if key not in self.idx:
            self.idx[key] = 2 + len(self.ins)
            self.ins.append('"%s"(%s)' % (cons, expr))
        return self.idx[key]

    def mac(self, x, xc, y, yc):
        ix = self.op(x, xc)
        iy = self.op(y, yc)
        self.body.append("v_mad_u64_u32 %%0, vcc, %%%d, %%%d, %%0" % (ix, iy))
        self.body.append("v_addc_co_u32 %1, vcc, 0, %1, vcc")

    def emit(self):
        if not self.body:
            return ""
        assert len(self.ins) + 2 <= 30, "asm operand limit"
        return '    asm("%s" : "+v"(lo), "+v"(hi) : %s : "vcc");\n' % (
            "\\n\\t".join(self.body), ", ".join(self.ins))


def gen_mul(N):
    out = []
    out.append("// ---- N = %d limbs ----\n" % N)
    out.append("template<> struct MontMulImpl<%d> {\n" % N)
    out.append("  template<class P>\n")
    out.append("  static __device__ __forceinline__ void mul(u32* __restrict__ r, const u32* a, const u32* b) {\n")
    out.append("    u64 lo = 0; u32 hi = 0; u32 m[%d]; u32 t[%d];\n" % (N, N))
    for k in range(2 * N):
        s = Stmt()
        for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
            s.mac("a[%d]" % i, "v", "b[%d]" % (k - i), "v")
        out.append(s.emit())
        s = Stmt()
        if k < N:
            for i in range(k):
                s.mac("m[%d]" % i, "v", "P::MOD[%d]" % (k - i), "s")
            out.append(s.emit())
            out.append("    m[%d] = (u32)lo * P::INV;\n" % k)
            s = Stmt()
            s.mac("m[%d]" % k, "v", "P::MOD[0]", "s")
            out.append(s.emit())
        else:
            for i in range(k - N + 1, N):
                s.mac("m[%d]" % i, "v", "P::MOD[%d]" % (k - i), "s")
            out.append(s.emit())
            out.append("    t[%d] = (u32)lo;\n" % (k - N))
        out.append("    lo = (lo >> 32) | ((u64)hi << 32); hi = 0;\n")
    out.append("    ff_final_sub<P, %d>(r, t, (u32)lo);\n" % N)
    out.append("  }\n};\n\n")
    return "".join(out)


if __name__ == "__main__":
    sys.stdout.write("// GENERATED FILE -- do not edit; regenerate with gen_mont.py\n")
    for n in (8, 12):
        sys.stdout.write(gen_mul(n))
