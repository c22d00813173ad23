#!/usr/bin/env python
"""Static check of the gfx950 assembly of the hot kernels for the one hazard hipcc does not pad
around inline asm: "VALU writes a VGPR, then a DPP instruction reads that VGPR as src0 within
2 wait states" (every instruction issued in between is one wait state; `s_nop N` is N+1).

The fused broadcast-FMA (`v_fmac_f64_dpp`) has no builtin, so mk_kernels.hip emits it as inline
asm WITHOUT leading `s_nop`s (a lone wavefront pays ~9 cycles per `s_nop 1`); this script is the
safety net, run by `__graft_entry__.build()` on the `-save-temps` assembly: it walks backwards from
every DPP instruction over all control-flow predecessors and fails the build if a producer of the
DPP source sits fewer than two wait states upstream.

usage: check_dpp_hazards.py file.s [kernel-name-substring ...]
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
NON_VALU_PREFIX = ("s_", "ds_", "global_", "buffer_", "scratch_", "flat_", ";")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_function(lines):
    """-> list of dicts(op, dst, src0, text, labels_before) and label -> index map."""
    insts, labels, pending = [], {}, []
    for raw in lines:
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\.L\w+):", line)
        if m:
            pending.append(m.group(1))
            continue
        if not line.startswith("\t") or line.strip().startswith("."):
            continue
        parts = line.strip().split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        for lb in pending:
            labels[lb] = len(insts)
        pending = []
        insts.append(dict(op=op, ops=ops, text=line.strip()))
    return insts, labels


def vgpr_written(inst):
    op = inst["op"]
    if op.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store", "v_cmp", "v_cmpx",
                      "v_readlane", "v_readfirstlane")):
        return set()
    if op.startswith(("ds_read", "global_load", "buffer_load", "scratch_load", "flat_load")):
        return set()  # memory returns are guarded by s_waitcnt, not by this hazard
    if not inst["ops"]:
        return set()
    if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")) and len(inst["ops"]) > 1:
        return regs_of(inst["ops"][0]) | regs_of(inst["ops"][1])  # both operands are written
    return regs_of(inst["ops"][0])


def wait_states(inst):
    if inst["op"] == "s_nop":
        return int(inst["ops"][0]) + 1
    return 1


def predecessors(insts, labels):
    """index -> list of predecessor instruction indices (fallthrough + branches)."""
    preds = {i: [] for i in range(len(insts))}
    for i, ins in enumerate(insts):
        op = ins["op"]
        falls = not (op == "s_branch" or op == "s_endpgm" or op == "s_setpc_b64")
        if falls and i + 1 < len(insts):
            preds[i + 1].append(i)
        if op.startswith(("s_cbranch", "s_branch")) and ins["ops"]:
            tgt = labels.get(ins["ops"][-1])
            if tgt is not None and tgt < len(insts):
                preds[tgt].append(i)
    return preds


def check_function(name, lines):
    insts, labels = parse_function(lines)
    preds = predecessors(insts, labels)
    errors, ndpp = [], 0
    for i, ins in enumerate(insts):
        if "row_newbcast" not in ins["text"] and "_dpp" not in ins["op"]:
            continue
        ndpp += 1
        # src0 of VOP1 (mov) is ops[1]; of VOP2 (fmac) is ops[1] as well (vdst, src0, src1)
        src = regs_of(ins["ops"][1].split(" ")[0]) if len(ins["ops"]) > 1 else set()
        # walk back up to 2 wait states along every path
        stack = [(p, 0) for p in preds[i]]
        seen = set()
        while stack:
            j, dist = stack.pop()
            if (j, dist) in seen:
                continue
            seen.add((j, dist))
            w = vgpr_written(insts[j])
            if w & src:
                errors.append("%s: '%s' reads %s written %d wait state(s) earlier by '%s'"
                              % (name, ins["text"], sorted(w & src), dist, insts[j]["text"]))
                continue
            nd = dist + wait_states(insts[j])
            if nd < 2:
                stack.extend((p, nd) for p in preds[j])
    return ndpp, errors


def main(argv):
    path = argv[1]
    want = argv[2:] or ["filter_kernel", "filter_obs_kernel", "filter_split_kernel", "smoother_", "adjoint_kernel", "loglik_sparse"]
    lines = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    total_err = []
    for k, (i, name) in enumerate(starts):
        if not any(w in name for w in want):
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = []
        for l in lines[i + 1:end]:
            body.append(l)
            if "s_endpgm" in l:
                break
        ndpp, errs = check_function(name, body)
        print("%-70s %4d DPP instructions, %d hazard(s)" % (name[:70], ndpp, len(errs)))
        total_err += errs
    for e in total_err[:40]:
        print("HAZARD", e)
    return 1 if total_err else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
