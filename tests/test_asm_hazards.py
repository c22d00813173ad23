"""scripts/check_asm_hazards.py (run by the Makefile and by the run-time shape compiler on the final gfx950 assembly):
the checker must SEE each hazard it names when one side sits in an inline-asm block, must accept the padded form, and
must leave pairs of two compiler instructions to the compiler.  Synthetic assembly, no GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "scripts", "check_asm_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def run(body, show_all=False):
    lines = ["\t" + ln if not ln.startswith((";", ".L")) else ln for ln in body.strip().split("\n")] + ["\ts_endpgm"]
    return chk.check_function("k", lines, show_all)[1]


MFMA = "v_mfma_f64_16x16x4_f64 v[0:7], v[8:9], v[10:11], v[0:7]"


@pytest.mark.parametrize("rule,producer,consumer,need", [
    ("M1", MFMA, "v_fmac_f64_dpp v[20:21], v[0:1], v[22:23] row_newbcast:3 row_mask:0xf bank_mask:0xf", 11),
    ("M1", "v_mfma_f64_4x4x4_4b_f64 v[0:1], v[8:9], v[10:11], v[0:1]", "v_add_f64 v[0:1], v[2:3], v[4:5]", 6),   # WAW
    ("M2", MFMA, "ds_write_b64 v30, v[6:7] offset:16", 18),
    ("T1", "v_rcp_f64_e32 v[2:3], v[4:5]", "v_fma_f64 v[6:7], -v[4:5], v[2:3], 1.0", 1),
    ("S1", "v_readlane_b32 s4, v1, 3", "v_readlane_b32 s6, v2, s4", 4),
    ("S2", "v_readfirstlane_b32 s8, v1", "global_load_dwordx2 v[2:3], v4, s[8:9]", 5),
])
def test_asm_side_is_flagged_until_padded(rule, producer, consumer, need):
    # consumer inside an asm block, directly behind the producer
    found = run("%s\n;;#ASMSTART\n%s\n;;#ASMEND" % (producer, consumer))
    assert len(found) == 1 and found[0].startswith(rule) and "%d required" % need in found[0], found
    # one wait state short
    if need > 1:
        assert run("%s\ns_nop %d\n;;#ASMSTART\n%s\n;;#ASMEND" % (producer, need - 2, consumer))
    # padded: s_nop N is N + 1 wait states (at most 8 per instruction)
    pad = "\n".join("s_nop %d" % (min(need - k, 8) - 1) for k in range(0, need, 8))
    assert run("%s\n%s\n;;#ASMSTART\n%s\n;;#ASMEND" % (producer, pad, consumer)) == []
    # both sides compiler code: not this script's business, unless asked
    assert run("%s\n%s" % (producer, consumer)) == []
    assert len(run("%s\n%s" % (producer, consumer), show_all=True)) == 1


def test_asm_valu_feeding_an_mfma_operand():
    asm = ";;#ASMSTART\nv_fmac_f64_dpp v[8:9], v[20:21], v[22:23] row_newbcast:1 row_mask:0xf bank_mask:0xf\n;;#ASMEND\n"
    assert run(asm + MFMA)[0].startswith("M3")
    assert run(asm + "s_nop 0\n" + MFMA)[0].startswith("M3")
    assert run(asm + "s_nop 1\n" + MFMA) == []
    assert run(asm + "v_mov_b32_e32 v40, v41\nv_mov_b32_e32 v42, v43\n" + MFMA) == []   # two independent instructions


def test_hazard_is_found_across_a_branch():
    body = MFMA + "\ns_cbranch_scc1 .LBB0_2\ns_nop 7\ns_nop 7\n.LBB0_2:\n;;#ASMSTART\nv_mov_b32_e32 v50, v3\n;;#ASMEND"
    found = run(body)
    assert len(found) == 1 and "1 wait state(s)" in found[0]     # the taken path: only the branch in between


def test_shipped_kernels_are_clean():
    """The assembly `make` kept of the three kernel files (present after __graft_entry__.build())."""
    build = os.path.join(ROOT, "build", "csrc")
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(build) for f in fs) if os.path.isdir(build) else []   # mk_wide.hip's slices sit in sub-directories
    files = [f for f in files if f.endswith("gfx950.s")]
    if not files:
        pytest.skip("no build/csrc assembly (run __graft_entry__.build())")
    for f in files:
        assert chk.main(["check_asm_hazards.py", f]) == 0, f
