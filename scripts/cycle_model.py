#!/usr/bin/env python
"""Dev helper: a first-order SIMD-time estimate of one kernel's main loop from its gfx950 assembly, to rank kernel variants
BEFORE spending GPU minutes on them.  Per-instruction SIMD costs are this repository's own MI355X measurements (DESIGN.md
section 4, profiles/HISTORY.md: every CU busy, two wavefronts per SIMD, 2.13 GHz under load): v_fmac_f64_dpp 6.3 cycles,
other f64 VALU 5.0, v_mfma_f64_16x16x4 64, v_mfma_f64_4x4x4_4b 16; 32-bit VALU and v_readlane 4 (issue rate of a 64-lane
instruction on a 16-lane pipe); s_nop N = N + 1.  LDS, scalar and memory instructions are counted but not priced (they issue
on other ports; their latency is what the partner wavefront hides or does not).  The estimate is a FLOOR of the issue-bound
time; calibration against measured launches is printed by --calibrate.

    python scripts/cycle_model.py file.s kernel_substring [models steps]      (default 4096 models x 2000 steps, 1024 SIMDs)
"""
import collections
import re
import sys

COST = {"dpp64": 6.3, "f64": 5.0, "mfma16": 64.0, "mfma4": 16.0, "valu32": 4.0, "trans64": 16.0}
CLOCK = 2.13e9
SIMDS = 1024


def main_loop(s, name):
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    best = None
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            span = k - labels[m.group(1)]
            if best is None or span > best[0]:
                best = (span, labels[m.group(1)], k)
    return body[best[1]:best[2]] if best else body


def classify(op, line):
    if op.startswith("v_mfma_f64_16x16"):
        return "mfma16"
    if op.startswith("v_mfma_f64_4x4"):
        return "mfma4"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "trans64"
    if op.startswith("v_") and ("_dpp" in op or "row_newbcast" in line) and "f64" in op:
        return "dpp64"
    if op.startswith("v_") and ("_f64" in op or op.startswith("v_mov_b64")):
        return "f64"
    if op.startswith("v_"):
        return "valu32"
    if op == "s_nop":
        return "nop"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def estimate(path, key):
    s = open(path).read()
    out = []
    for name in [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m]:
        c = collections.Counter()
        nop = 0
        for l in main_loop(s, name):
            l = l.split(";")[0].strip()
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            k = classify(op, l)
            c[k] += 1
            if k == "nop":
                nop += int(l.split()[1]) + 1
        cycles = sum(COST[k] * v for k, v in c.items() if k in COST) + nop
        out.append((name, c, nop, cycles))
    return out


if __name__ == "__main__":
    path, key = sys.argv[1], sys.argv[2]
    models = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
    for name, c, nop, cycles in estimate(path, key):
        ms = cycles * models * steps / SIMDS / CLOCK * 1e3
        print(name)
        print("  " + ", ".join("%s %d" % kv for kv in sorted(c.items())) + ", s_nop wait states %d" % nop)
        print("  priced issue cycles per wavefront-iteration: %.0f  ->  %.1f ms for %d models x %d steps (floor; x 1.05-1.10 measured)"
              % (cycles, ms, models, steps))
