#!/usr/bin/env python3
"""Generate dpp_blocks_gen.hpp: inline-asm building blocks around
`v_fmac_f64_dpp ... row_newbcast:k` for the 16-lanes-per-problem Riccati kernel.

hipcc (ROCm 7.2) never folds a 64-bit `v_mov_b64_dpp` into the consuming
`v_fma_f64` (the DPP combiner runs before FMA->FMAC conversion and V_FMA_F64 is
VOP3-only on gfx950), and an asm statement must be a string literal, so the
fused broadcast-FMA blocks are generated here, one specialisation per block
length R = 1..16.

Blocks (all lanes of the wave must be active; a "row" is 16 consecutive lanes):

  rank1<R,K,NEG,WAIT>(C, A, b)   C[i] (+/-)= A[i]@lane K of the row * b, i < R
  spread<R,NEG,WAIT>(C, a, b)    C[i] (+/-)= a@lane i of the row * b,    i < R
  dotv<R,WAIT>(acc, x, B)        acc[k % 4] += x@lane k of the row * B[k], k < R
  spreadv<R,WAIT>(C, a, B)       C[i] += a@lane i of the row * B[i],     i < R

WAIT prepends `s_nop 1`: the two wait states gfx9 needs between a VALU write of
a VGPR and a DPP read of it (the compiler pads its own instructions but not
what is inside an asm string).
"""
import sys

MAXR = 16
DPP = "row_newbcast:%{k} row_mask:0xf bank_mask:0xf"


def emit_rank1(out, R):
    # operands: %0..%R-1 = C (rw), %R..%2R-1 = A, %2R = b, %2R+1 = K
    def body(neg):
        lines = []
        for i in range(R):
            b = ("-" if neg else "") + "%" + str(2 * R)
            lines.append(
                f"v_fmac_f64_dpp %{i}, %{R + i}, {b} row_newbcast:%{2 * R + 1} "
                "row_mask:0xf bank_mask:0xf")
        return lines

    outs = ", ".join(f'"+v"(C[{i}])' for i in range(R))
    ins = ", ".join(f'"v"(A[{i}])' for i in range(R)) + ', "v"(b), "n"(K)'
    out.append(f"template <int K, bool NEG, bool WAIT>\n"
               f"__device__ __forceinline__ void rank1_{R}(double *C, const double *A, double b) {{")
    for neg in (False, True):
        for wait in (False, True):
            lines = (["s_nop 1"] if wait else []) + body(neg)
            s = "\\n\\t".join(lines)
            out.append(f"  if constexpr (NEG == {str(neg).lower()} && WAIT == {str(wait).lower()})\n"
                       f'    asm volatile("{s}"\n        : {outs}\n        : {ins});')
    out.append("}\n")


def emit_spread(out, R):
    # operands: %0..%R-1 = C (rw), %R = a, %R+1 = b ; lane i literal
    outs = ", ".join(f'"+v"(C[{i}])' for i in range(R))
    ins = '"v"(a), "v"(b)'
    out.append(f"template <bool NEG, bool WAIT>\n"
               f"__device__ __forceinline__ void spread_{R}(double *C, double a, double b) {{")
    for neg in (False, True):
        for wait in (False, True):
            lines = ["s_nop 1"] if wait else []
            for i in range(R):
                b = ("-" if neg else "") + "%" + str(R + 1)
                lines.append(f"v_fmac_f64_dpp %{i}, %{R}, {b} row_newbcast:{i} "
                             "row_mask:0xf bank_mask:0xf")
            s = "\\n\\t".join(lines)
            out.append(f"  if constexpr (NEG == {str(neg).lower()} && WAIT == {str(wait).lower()})\n"
                       f'    asm volatile("{s}"\n        : {outs}\n        : {ins});')
    out.append("}\n")


def emit_dotv(out, R):
    # operands: %0..%3 = acc (rw), %4 = x, %5.. = B[k]
    outs = ", ".join(f'"+v"(acc[{i}])' for i in range(4))
    ins = '"v"(x), ' + ", ".join(f'"v"(B[{k}])' for k in range(R))
    out.append(f"template <bool WAIT>\n"
               f"__device__ __forceinline__ void dotv_{R}(double *acc, double x, const double *B) {{")
    for wait in (False, True):
        lines = ["s_nop 1"] if wait else []
        for k in range(R):
            lines.append(f"v_fmac_f64_dpp %{k % 4}, %4, %{5 + k} row_newbcast:{k} "
                         "row_mask:0xf bank_mask:0xf")
        s = "\\n\\t".join(lines)
        out.append(f"  if constexpr (WAIT == {str(wait).lower()})\n"
                   f'    asm volatile("{s}"\n        : {outs}\n        : {ins});')
    out.append("}\n")


def emit_spreadv(out, R):
    # operands: %0..%R-1 = C (rw), %R = a, %R+1.. = B[i]
    outs = ", ".join(f'"+v"(C[{i}])' for i in range(R))
    ins = '"v"(a), ' + ", ".join(f'"v"(B[{i}])' for i in range(R))
    out.append(f"template <bool WAIT>\n"
               f"__device__ __forceinline__ void spreadv_{R}(double *C, double a, const double *B) {{")
    for wait in (False, True):
        lines = ["s_nop 1"] if wait else []
        for i in range(R):
            lines.append(f"v_fmac_f64_dpp %{i}, %{R}, %{R + 1 + i} row_newbcast:{i} "
                         "row_mask:0xf bank_mask:0xf")
        s = "\\n\\t".join(lines)
        out.append(f"  if constexpr (WAIT == {str(wait).lower()})\n"
                   f'    asm volatile("{s}"\n        : {outs}\n        : {ins});')
    out.append("}\n")


def emit_x(out, kind, R, K):
    """Whole products in one asm statement (fewer statement boundaries for the
    compiler to pad): rank1x: C[i] += A[i]@lane k * B[k]; spreadx: C[i] += A[k]@lane i * B[k]."""
    outs = ", ".join(f'"+v"(C[{i}])' for i in range(R))
    na = R if kind == "rank1x" else K
    ins = ", ".join(f'"v"(A[{i}])' for i in range(na)) + ", " + ", ".join(f'"v"(B[{k}])' for k in range(K))
    out.append(f"template <bool WAIT>\n"
               f"__device__ __forceinline__ void {kind}_{R}_{K}(double *C, const double *A, const double *B) {{")
    for wait in (False, True):
        lines = ["s_nop 1"] if wait else []
        for k in range(K):
            for i in range(R):
                if kind == "rank1x":
                    lines.append(f"v_fmac_f64_dpp %{i}, %{R + i}, %{R + na + k} row_newbcast:{k} "
                                 "row_mask:0xf bank_mask:0xf")
                else:
                    lines.append(f"v_fmac_f64_dpp %{i}, %{R + k}, %{R + na + k} row_newbcast:{i} "
                                 "row_mask:0xf bank_mask:0xf")
        s = "\\n\\t".join(lines)
        out.append(f"  if constexpr (WAIT == {str(wait).lower()})\n"
                   f'    asm volatile("{s}"\n        : {outs}\n        : {ins});')
    out.append("}\n")


XSIZES = list(range(1, 17))


def main(path):
    out = [
        "// GENERATED by gen_dpp_blocks.py -- do not edit.",
        "// Fused broadcast-FMA blocks on v_fmac_f64_dpp row_newbcast (gfx950).",
        "#pragma once",
        "#include <hip/hip_runtime.h>",
        "namespace sipamd {",
        "namespace dppgen {",
        "",
    ]
    for R in range(1, MAXR + 1):
        emit_rank1(out, R)
        emit_spread(out, R)
        emit_dotv(out, R)
        emit_spreadv(out, R)
    for R in XSIZES:
        for K in XSIZES:
            emit_x(out, "rank1x", R, K)
            emit_x(out, "spreadx", R, K)
    out.append("} // namespace dppgen\n")
    for kind in ("rank1x", "spreadx"):
        out.append(f"template <int R, int K, bool WAIT>\n"
                   f"__device__ __forceinline__ void {kind}(double *C, const double *A, const double *B) {{")
        conds = " || ".join(f"V == {v}" for v in XSIZES)
        out.append("  constexpr auto ok = [](int V) { return %s; };" % conds)
        out.append("  static_assert(ok(R) && ok(K), \"add the size to XSIZES in gen_dpp_blocks.py\");")
        for R in XSIZES:
            for K in XSIZES:
                out.append(f"  if constexpr (R == {R} && K == {K}) dppgen::{kind}_{R}_{K}<WAIT>(C, A, B);")
        out.append("}\n")
    # dispatchers on R
    out.append("template <int R, int K, bool NEG, bool WAIT>\n"
               "__device__ __forceinline__ void rank1(double *C, const double *A, double b) {")
    out.append("  static_assert(R >= 0 && R <= %d && K >= 0 && K < 16);" % MAXR)
    for R in range(1, MAXR + 1):
        out.append(f"  if constexpr (R == {R}) dppgen::rank1_{R}<K, NEG, WAIT>(C, A, b);")
    out.append("}\n")
    out.append("template <int R, bool NEG, bool WAIT>\n"
               "__device__ __forceinline__ void spread(double *C, double a, double b) {")
    out.append("  static_assert(R >= 0 && R <= %d);" % MAXR)
    for R in range(1, MAXR + 1):
        out.append(f"  if constexpr (R == {R}) dppgen::spread_{R}<NEG, WAIT>(C, a, b);")
    out.append("}\n")
    out.append("template <int R, bool WAIT>\n"
               "__device__ __forceinline__ void dotv(double *acc, double x, const double *B) {")
    out.append("  static_assert(R >= 0 && R <= %d);" % MAXR)
    for R in range(1, MAXR + 1):
        out.append(f"  if constexpr (R == {R}) dppgen::dotv_{R}<WAIT>(acc, x, B);")
    out.append("}\n")
    out.append("template <int R, bool WAIT>\n"
               "__device__ __forceinline__ void spreadv(double *C, double a, const double *B) {")
    out.append("  static_assert(R >= 0 && R <= %d);" % MAXR)
    for R in range(1, MAXR + 1):
        out.append(f"  if constexpr (R == {R}) dppgen::spreadv_{R}<WAIT>(C, a, B);")
    out.append("}\n")
    out.append("} // namespace sipamd")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "dpp_blocks_gen.hpp")
