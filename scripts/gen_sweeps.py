#!/usr/bin/env python3
"""Writes metran_amd/csrc/mk_sweeps.h: the DPP broadcast-FMA sweeps of the 16-lane smoother as ONE asm statement each.

Why: hipcc's hazard recogniser counts an inline-asm statement as zero wait states and assumes every asm result needs
one before another asm statement may touch it, so a sweep written as one statement per v_fmac_f64_dpp gets an
`s_nop 0` at every sweep boundary (the next sweep's first FMA accumulates into a register the previous sweep's
statements wrote): 31 per step of smoother_record_kernel<8,2>, each a full issue slot.  A statement may have at
most 30 operands ("+v" counts twice): for n <= 10 every sweep is ONE statement; for 11 <= n <= 16 (round 3) a sweep is
TILED into statements of <= 30 operands -- blocks of 7 broadcast sources x 7 or 8 accumulators, ordered so that every
value is final before it is read -- 3 to 6 statements (= padding nops) per sweep instead of one per FMA."""
import os

DPP = " row_mask:0xf bank_mask:0xf"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metran_amd", "csrc", "mk_sweeps.h")


def stmt(lines, outs, ins, indent="        "):
    body = "\n".join('%s    "%s\\n\\t"' % (indent, l) for l in lines)
    return "%sasm volatile(\n%s\n%s    : %s\n%s    : %s);\n" % (indent, body, indent, ", ".join(outs), indent, ", ".join(ins) if ins else "")


def gen(n):
    s = "template <>\nstruct Sweeps<%d> {\n    static constexpr bool fused = true;\n" % n
    # forward substitution: z[c] -= bcast_c(A[k]) z[k], k = 0..n-2, c = k+1..n-1     (operands: z 0..n-1, A n..2n-2)
    lines = ["v_fmac_f64_dpp %%%d, %%%d, -%%%d row_newbcast:%d%s" % (c, n + k, k, c, DPP) for k in range(n - 1) for c in range(k + 1, n)]
    s += "    // L y = b: z[c] -= L(c,k) y_k, L(c,k) = A[k] of lane c\n"
    s += "    static __device__ __forceinline__ void forward(double (&z)[%d], const double (&A)[%d])\n    {\n" % (n, n)
    s += stmt(lines, ['"+v"(z[%d])' % c for c in range(n)], ['"v"(A[%d])' % k for k in range(n - 1)]) if lines else ""
    s += "    }\n"
    # backward substitution: z[c] -= bcast_k(A[c]) z[k], k = n-1..1, c = 0..k-1
    lines = ["v_fmac_f64_dpp %%%d, %%%d, -%%%d row_newbcast:%d%s" % (c, n + c, k, k, DPP) for k in range(n - 1, 0, -1) for c in range(k)]
    s += "    // L^T x = y: z[c] -= L(k,c) z_k, L(k,c) = A[c] of lane k\n"
    s += "    static __device__ __forceinline__ void backward(double (&z)[%d], const double (&A)[%d])\n    {\n" % (n, n)
    s += stmt(lines, ['"+v"(z[%d])' % c for c in range(n)], ['"v"(A[%d])' % c for c in range(n - 1)]) if lines else ""
    s += "    }\n"
    # smoothed mean: acc0 / acc1 += bcast_c(delta) z[c]           (operands: acc0 0, acc1 1, delta 2, z 3..)
    lines = ["v_fmac_f64_dpp %%%d, %%2, %%%d row_newbcast:%d%s" % (c % 2, 3 + c, c, DPP) for c in range(n)]
    s += "    // acc0 + acc1 += sum_c bcast_c(delta) z[c] (two chains)\n"
    s += "    static __device__ __forceinline__ void mean(double &acc0, double &acc1, double delta, const double (&z)[%d])\n    {\n" % n
    s += stmt(lines, ['"+v"(acc0)', '"+v"(acc1)'], ['"v"(delta)'] + ['"v"(z[%d])' % c for c in range(n)])
    s += "    }\n"
    # V = J D: V[c] = sum_k bcast_k(D[c]) z[k]                    (operands: V 0..n-1 (early clobber), D n.., z 2n..)
    lines = ["v_mov_b64 %%%d, 0" % c for c in range(n)]
    lines += ["v_fmac_f64_dpp %%%d, %%%d, %%%d row_newbcast:%d%s" % (c, n + c, 2 * n + k, k, DPP) for k in range(n) for c in range(n)]
    s += "    // V = J D (row r): V[c] = sum_k J[r][k] D[k][c], D[k][:] broadcast from lane k\n"
    s += "    static __device__ __forceinline__ void jd(double (&V)[%d], const double (&D)[%d], const double (&z)[%d])\n    {\n" % (n, n, n)
    s += stmt(lines, ['"=&v"(V[%d])' % c for c in range(n)], ['"v"(D[%d])' % c for c in range(n)] + ['"v"(z[%d])' % k for k in range(n)])
    s += "    }\n"
    # Ps += V J^T: P[c] += bcast_c(z[k]) V[k], in chunks of k      (operands: P 0..n-1, then (z[k], V[k]) pairs)
    m = max(1, (30 - 2 * n) // 2)
    s += "    // Ps[r][c] += sum_k V[r][k] J[c][k], J[c][k] broadcast from lane c\n"
    s += "    static __device__ __forceinline__ void vjt(double (&P)[%d], const double (&z)[%d], const double (&V)[%d])\n    {\n" % (n, n, n)
    for k0 in range(0, n, m):
        ks = list(range(k0, min(n, k0 + m)))
        lines = ["v_fmac_f64_dpp %%%d, %%%d, %%%d row_newbcast:%d%s" % (c, n + 2 * i, n + 2 * i + 1, c, DPP) for i, k in enumerate(ks) for c in range(n)]
        ins = []
        for k in ks:
            ins += ['"v"(z[%d])' % k, '"v"(V[%d])' % k]
        s += stmt(lines, ['"+v"(P[%d])' % c for c in range(n)], ins)
    s += "    }\n};\n\n"
    return s


def blocks(n, size):
    return [list(range(i, min(n, i + size))) for i in range(0, n, size)]


def gen_tiled(n):
    """11 <= n <= 16: the same five sweeps, each as a short sequence of <= 30-operand statements."""
    s = "template <>\nstruct Sweeps<%d> {\n    static constexpr bool fused = true;\n" % n
    KB = blocks(n, 7)

    def tri(forward):
        out = ""
        order = KB if forward else KB[::-1]
        for kb in order:
            # diagonal tile: sources and accumulators inside the block
            if forward:
                pairs = [(k, c) for k in kb for c in kb if c > k]            # z[c] -= bcast_c(A[k]) z[k]
            else:
                pairs = [(k, c) for k in kb[::-1] for c in kb if c < k]      # z[c] -= bcast_k(A[c]) z[k]
            if pairs:
                zs = sorted({c for _, c in pairs} | {k for k, _ in pairs})
                As = sorted({(k if forward else c) for k, c in pairs})
                zi = {c: i for i, c in enumerate(zs)}
                ai = {k: len(zs) + i for i, k in enumerate(As)}
                lines = ["v_fmac_f64_dpp %%%d, %%%d, -%%%d row_newbcast:%d%s" % (zi[c], ai[k if forward else c], zi[k], c if forward else k, DPP)
                         for k, c in pairs]
                assert 2 * len(zs) + len(As) <= 30
                out += stmt(lines, ['"+v"(z[%d])' % c for c in zs], ['"v"(A[%d])' % k for k in As])
            # off-diagonal tiles: this block's (final) sources into the accumulators of the blocks still to come
            rest = [c for c in range(n) if (c > kb[-1] if forward else c < kb[0])]
            for cb in blocks(len(rest), 8):
                cs = [rest[i] for i in cb]
                pairs = [(k, c) for k in (kb if forward else kb[::-1]) for c in cs]
                zi = {c: i for i, c in enumerate(cs)}
                ki = {k: len(cs) + i for i, k in enumerate(kb)}
                As = sorted({(k if forward else c) for k, c in pairs})
                ai = {a_: len(cs) + len(kb) + i for i, a_ in enumerate(As)}
                lines = ["v_fmac_f64_dpp %%%d, %%%d, -%%%d row_newbcast:%d%s" % (zi[c], ai[k if forward else c], ki[k], c if forward else k, DPP)
                         for k, c in pairs]
                assert 2 * len(cs) + len(kb) + len(As) <= 30, (n, len(cs), len(kb), len(As))
                out += stmt(lines, ['"+v"(z[%d])' % c for c in cs], ['"v"(z[%d])' % k for k in kb] + ['"v"(A[%d])' % a_ for a_ in As])
        return out

    s += "    // L y = b: z[c] -= L(c,k) y_k, L(c,k) = A[k] of lane c\n"
    s += "    static __device__ __forceinline__ void forward(double (&z)[%d], const double (&A)[%d])\n    {\n" % (n, n)
    s += tri(True) + "    }\n"
    s += "    // L^T x = y: z[c] -= L(k,c) z_k, L(k,c) = A[c] of lane k\n"
    s += "    static __device__ __forceinline__ void backward(double (&z)[%d], const double (&A)[%d])\n    {\n" % (n, n)
    s += tri(False) + "    }\n"
    lines = ["v_fmac_f64_dpp %%%d, %%2, %%%d row_newbcast:%d%s" % (c % 2, 3 + c, c, DPP) for c in range(n)]
    s += "    // acc0 + acc1 += sum_c bcast_c(delta) z[c] (two chains)\n"
    s += "    static __device__ __forceinline__ void mean(double &acc0, double &acc1, double delta, const double (&z)[%d])\n    {\n" % n
    s += stmt(lines, ['"+v"(acc0)', '"+v"(acc1)'], ['"v"(delta)'] + ['"v"(z[%d])' % c for c in range(n)])
    s += "    }\n"
    # V = J D: tiles of 8 accumulators x 6 sources (first tile of a column block zeroes its accumulators)
    s += "    // V = J D (row r): V[c] = sum_k J[r][k] D[k][c], D[k][:] broadcast from lane k\n"
    s += "    static __device__ __forceinline__ void jd(double (&V)[%d], const double (&D)[%d], const double (&z)[%d])\n    {\n" % (n, n, n)
    for cs in blocks(n, 8):
        for bi, ks in enumerate(blocks(n, 6)):
            vi = {c: i for i, c in enumerate(cs)}
            di = {c: len(cs) + i for i, c in enumerate(cs)}
            zi = {k: 2 * len(cs) + i for i, k in enumerate(ks)}
            lines = (["v_mov_b64 %%%d, 0" % vi[c] for c in cs] if bi == 0 else [])
            lines += ["v_fmac_f64_dpp %%%d, %%%d, %%%d row_newbcast:%d%s" % (vi[c], di[c], zi[k], k, DPP) for k in ks for c in cs]
            outs = ['"%s"(V[%d])' % ("=&v" if bi == 0 else "+v", c) for c in cs]
            assert (1 if bi == 0 else 2) * len(cs) + len(cs) + len(ks) <= 30
            s += stmt(lines, outs, ['"v"(D[%d])' % c for c in cs] + ['"v"(z[%d])' % k for k in ks])
    s += "    }\n"
    s += "    // Ps[r][c] += sum_k V[r][k] J[c][k], J[c][k] broadcast from lane c\n"
    s += "    static __device__ __forceinline__ void vjt(double (&P)[%d], const double (&z)[%d], const double (&V)[%d])\n    {\n" % (n, n, n)
    for cs in blocks(n, 8):
        for ks in blocks(n, 7):
            pi = {c: i for i, c in enumerate(cs)}
            lines = ["v_fmac_f64_dpp %%%d, %%%d, %%%d row_newbcast:%d%s" % (pi[c], len(cs) + 2 * i, len(cs) + 2 * i + 1, c, DPP)
                     for i, k in enumerate(ks) for c in cs]
            ins = []
            for k in ks:
                ins += ['"v"(z[%d])' % k, '"v"(V[%d])' % k]
            assert 2 * len(cs) + 2 * len(ks) <= 30
            s += stmt(lines, ['"+v"(P[%d])' % c for c in cs], ins)
    s += "    }\n};\n\n"
    return s


def render():
    h = ("// mk_sweeps.h -- GENERATED by scripts/gen_sweeps.py (do not edit): the broadcast-FMA sweeps of the 16-lane smoother,\n"
         "// one asm statement per sweep (see the generator for why).  Included by mk_prims.h inside namespace mk.\n"
         "// The DPP source operands of every statement (A, D, delta, z of the LAST sweep) are old values: the callers keep\n"
         "// the two wait states between their producers and the statement (dpp_guard), scripts/check_dpp_hazards.py checks.\n"
         "#pragma once\n\n"
         "template <int n>\nstruct Sweeps {\n    static constexpr bool fused = false; // wider groups: the per-FMA primitives of Group<16>\n};\n\n")
    for n in range(2, 11):
        h += gen(n)
    h += ("// 11 <= n <= 16: a statement takes at most 30 operands -- every sweep is tiled into a few statements (blocks of 7\n"
          "// broadcast sources x 7 or 8 accumulators; a block's sources are final before any later tile reads them)\n"
          "#ifndef MK_NO_TILED_SWEEPS\n")
    for n in range(11, 17):
        h += gen_tiled(n)
    h += "#endif\n"
    return h


def main():
    open(OUT, "w").write(render())
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
