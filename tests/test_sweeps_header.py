"""The GENERATED csrc/mk_sweeps.h (scripts/gen_sweeps.py): every asm statement of every sweep is parsed from the header text
and emulated on random data -- forward / backward substitution, the smoothed mean, V = J D and Ps += V J^T -- for every
state dimension it covers (2 .. 16; 11 .. 16 are TILED into several <= 30-operand statements, round 3).  Catches an
ordering or operand-index mistake of the generator without a GPU; the GPU parity tests then check the kernels."""
import os
import re

import numpy as np

from conftest import ROOT

SRC = open(os.path.join(ROOT, "metran_amd", "csrc", "mk_sweeps.h")).read()


def _parse(n, fn):
    a = SRC.index("struct Sweeps<%d>" % n)
    body = SRC[a:SRC.index("};", a)]
    f = body.index("void %s(" % fn)
    stmts = body[f:body.index("\n    }\n", f)].split("asm volatile(")[1:]
    ops = []
    for st in stmts:
        lines = re.findall(r'"(v_\w+ [^"]*?)\\n\\t"', st)
        operands = re.findall(r'"[=+&]*v"\(([\w\[\]]+)\)', st[st.index(":"):])
        assert len(operands) + len(re.findall(r'"\+v"', st)) <= 30, (n, fn)   # the inline-asm operand limit
        for ln in lines:
            m = re.match(r"v_fmac_f64_dpp %(\d+), %(\d+), (-?)%(\d+) row_newbcast:(\d+)", ln)
            if m:
                d, s0, neg, s1, bc = m.groups()
                ops.append(("fmac", operands[int(d)], operands[int(s0)], -1.0 if neg else 1.0, operands[int(s1)], int(bc)))
            else:
                ops.append(("zero", operands[int(re.match(r"v_mov_b64 %(\d+), 0", ln).group(1))]))
    return ops, len(stmts)


def _i(name):
    return int(re.match(r"\w+\[(\d+)\]", name).group(1))


def test_every_generated_sweep_computes_what_it_says():
    rng = np.random.default_rng(0)
    statements = {}
    for n in range(2, 17):
        L = np.tril(rng.standard_normal((n, n)), -1)      # lane c holds L(c, k) in A[k]
        z = rng.standard_normal((n, n))                   # lane r: row r of the right-hand sides
        ref = z.copy()
        for k in range(n - 1):
            for c in range(k + 1, n):
                ref[:, c] -= L[c, k] * ref[:, k]
        got = z.copy()
        ops, ns = _parse(n, "forward")
        for _, d, s0, sg, s1, bc in ops:
            got[:, _i(d)] += sg * L[bc, _i(s0)] * got[:, _i(s1)]
        np.testing.assert_allclose(got, ref, atol=1e-12)
        ref = z.copy()
        for k in range(n - 1, 0, -1):
            for c in range(k):
                ref[:, c] -= L[k, c] * ref[:, k]
        got = z.copy()
        for _, d, s0, sg, s1, bc in _parse(n, "backward")[0]:
            got[:, _i(d)] += sg * L[bc, _i(s0)] * got[:, _i(s1)]
        np.testing.assert_allclose(got, ref, atol=1e-12)
        D, J = rng.standard_normal((n, n)), rng.standard_normal((n, n))
        V = np.full((n, n), np.nan)
        for op in _parse(n, "jd")[0]:
            if op[0] == "zero":
                V[:, _i(op[1])] = 0.0
            else:
                V[:, _i(op[1])] += D[op[5], _i(op[2])] * J[:, _i(op[4])]
        np.testing.assert_allclose(V, J @ D, atol=1e-12)
        P = rng.standard_normal((n, n))
        ref = P + V @ J.T
        for _, d, s0, sg, s1, bc in _parse(n, "vjt")[0]:
            P[:, _i(d)] += J[bc, _i(s0)] * V[:, _i(s1)]
        np.testing.assert_allclose(P, ref, atol=1e-12)
        delta, acc = rng.standard_normal(n), np.zeros(n)
        for _, d, s0, sg, s1, bc in _parse(n, "mean")[0]:
            acc += delta[bc] * J[:, _i(s1)]
        np.testing.assert_allclose(acc, J @ delta, atol=1e-12)
        statements[n] = ns
    assert all(statements[n] == 1 for n in range(2, 11)) and all(2 <= statements[n] <= 6 for n in range(11, 17))


def test_header_is_what_the_generator_writes():
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_sweeps", os.path.join(ROOT, "scripts", "gen_sweeps.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert g.render() == SRC
