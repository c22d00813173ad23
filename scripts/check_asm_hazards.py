#!/usr/bin/env python
"""Static check of the gfx950 assembly for the software-managed hazards that involve INLINE ASM (companion of
check_dpp_hazards.py, which covers "VALU write -> DPP read").

hipcc's hazard recogniser pads compiler-emitted instruction pairs; an inline-asm statement is opaque to it (it is not
classified as VALU / MFMA / LDS), so a pair with one side inside ``;;#ASMSTART ... ;;#ASMEND`` gets no wait states.  The
kernels mix builtin MFMA (``__builtin_amdgcn_mfma_f64_16x16x4f64``), compiler VALU code and inline-asm
``v_fmac_f64_dpp`` / ``v_readlane`` / exec-masked ``ds_write``: this script walks the final assembly and reports every
pair below that is closer than the required number of wait states AND has at least one side inside an asm block
(pairs of two compiler instructions are the compiler's business; ``--all`` lists those too, as a check of the table:
it must print none).  Wait states as LLVM counts them: every issued instruction is one, ``s_nop N`` is N + 1.

  rule                                                              wait states   (LLVM GCNHazardRecognizer, gfx90a/gfx940)
  M1  v_mfma_f64_16x16x4 writes VGPR -> VALU reads/writes it              11
      v_mfma_f64_4x4x4   writes VGPR -> VALU reads/writes it               6
  M2  v_mfma_f64_16x16x4 writes VGPR -> LDS/VMEM reads it (store data)    18
      v_mfma_f64_4x4x4   writes VGPR -> LDS/VMEM reads it                  9
  M3  VALU writes VGPR -> v_mfma_f64 reads it (SrcA/B/C)                    2
  S1  VALU writes SGPR -> v_readlane / v_writelane lane select              4
  S2  VALU writes SGPR -> VMEM reads that SGPR                              5
  T1  transcendental VALU (v_rcp/rsq/sqrt/exp/log/sin/cos) writes VGPR
      -> non-transcendental VALU reads it                                   1
  E1  VALU writes EXEC (v_cmpx) -> DPP instruction                          5

usage: check_asm_hazards.py file.s [--all] [kernel-name-substring ...]
"""
import re
import sys

VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
MEM = ("ds_", "global_", "buffer_", "scratch_", "flat_")
MAXWS = 18


def regs(tok, rx):
    out = set()
    for m in rx.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_function(lines):
    insts, labels, pending, in_asm = [], {}, [], False
    for raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
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
        insts.append(dict(op=op, ops=ops, text=line.strip(), asm=in_asm))
    return insts, labels


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_mfma")


def vgpr_writes(ins):
    op = ins["op"]
    if not op.startswith("v_") or op.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    if not ins["ops"]:
        return set()
    w = regs(ins["ops"][0], VREG)
    if op.startswith("v_swap"):
        w |= regs(ins["ops"][1], VREG)
    return w


def vgpr_reads(ins):
    """every VGPR an instruction reads (for a VALU: all operands but the first, plus the first when the opcode
    accumulates into it; for stores / LDS writes: every operand)"""
    op, ops = ins["op"], ins["ops"]
    if not ops:
        return set()
    if op.startswith("v_"):
        r = set()
        for o in ops[1:]:
            r |= regs(o.split(" ")[0], VREG)
        if op.startswith(("v_fmac", "v_mac", "v_swap", "v_writelane", "v_cmp", "v_cmpx", "v_dot")) or "_dpp" in op:
            r |= regs(ops[0], VREG)      # dst is also a source (DPP: bound_ctrl-less lanes keep the old value)
        return r
    if op.startswith(MEM):
        r = set()
        start = 1 if op.startswith(("ds_read", "global_load", "buffer_load", "scratch_load", "flat_load")) else 0
        for o in ops[start:]:
            r |= regs(o.split(" ")[0], VREG)
        return r
    return set()


def sgpr_writes_by_valu(ins):
    op = ins["op"]
    if op.startswith(("v_readlane", "v_readfirstlane")) and ins["ops"]:
        return regs(ins["ops"][0], SREG)
    if op.startswith("v_cmp") and ins["ops"] and not op.startswith("v_cmpx"):
        return regs(ins["ops"][0], SREG)
    return set()


def wait_states(ins):
    if ins["op"] == "s_nop":
        return int(ins["ops"][0]) + 1
    return 1


def successors(insts, labels):
    succ = {i: [] for i in range(len(insts))}
    for i, ins in enumerate(insts):
        op = ins["op"]
        falls = not (op == "s_branch" or op == "s_endpgm" or op == "s_setpc_b64")
        if falls and i + 1 < len(insts):
            succ[i].append(i + 1)
        if op.startswith(("s_cbranch", "s_branch")) and ins["ops"]:
            tgt = labels.get(ins["ops"][-1])
            if tgt is not None and tgt < len(insts):
                succ[i].append(tgt)
    return succ


def rules_for_producer(ins):
    """-> list of (rule, regs, kind, required, consumer predicate, consumer register extractor)"""
    op = ins["op"]
    out = []
    if op.startswith("v_mfma_f64"):
        big = "16x16" in op
        d = regs(ins["ops"][0], VREG)
        out.append(("M1", d, "v", 11 if big else 6, lambda c: is_valu(c["op"]), lambda c: vgpr_reads(c) | vgpr_writes(c)))
        out.append(("M2", d, "v", 18 if big else 9, lambda c: c["op"].startswith(MEM), vgpr_reads))
    elif is_valu(op):
        w = vgpr_writes(ins)
        if w:
            out.append(("M3", w, "v", 2, lambda c: c["op"].startswith("v_mfma_f64"),
                        lambda c: set().union(*[regs(o, VREG) for o in c["ops"][1:]])))
            if op.startswith(TRANS):
                out.append(("T1", w, "v", 1, lambda c: is_valu(c["op"]) and not c["op"].startswith(TRANS), vgpr_reads))
        s = sgpr_writes_by_valu(ins)
        if s:
            out.append(("S1", s, "s", 4, lambda c: c["op"].startswith(("v_readlane", "v_writelane")),
                        lambda c: regs(c["ops"][2], SREG) if len(c["ops"]) > 2 else set()))
            out.append(("S2", s, "s", 5, lambda c: c["op"].startswith(("global_", "buffer_", "scratch_", "flat_")),
                        lambda c: set().union(*[regs(o, SREG) for o in c["ops"]])))
        if op.startswith("v_cmpx"):
            out.append(("E1", {-1}, "e", 5, lambda c: "_dpp" in c["op"] or "row_newbcast" in c["text"], lambda c: {-1}))
    return out


def check_function(name, lines, show_all):
    insts, labels = parse_function(lines)
    succ = successors(insts, labels)
    found, counts = [], {}
    for i, ins in enumerate(insts):
        for rule, rset, kind, need, is_consumer, cregs in rules_for_producer(ins):
            counts[rule] = counts.get(rule, 0) + 1
            stack = [(j, 0) for j in succ[i]]
            seen = set()
            while stack:
                j, dist = stack.pop()
                if dist >= need or (j, dist) in seen:
                    continue
                seen.add((j, dist))
                c = insts[j]
                if is_consumer(c) and (cregs(c) & rset) and (show_all or ins["asm"] or c["asm"]):
                    found.append("%s %s: '%s'%s -> '%s'%s: %d wait state(s), %d required"
                                 % (rule, name[:60], ins["text"], " [asm]" if ins["asm"] else "", c["text"],
                                    " [asm]" if c["asm"] else "", dist, need))
                stack.extend((k, dist + wait_states(c)) for k in succ[j])
    return counts, found


def main(argv):
    args = [a for a in argv[1:] if a != "--all"]
    show_all = "--all" in argv
    path = args[0]
    want = args[1:] or [""]
    lines = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    total = []
    for k, (i, name) in enumerate(starts):
        if not any(w in name for w in want):
            continue
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = []
        for l in lines[i + 1:end]:
            body.append(l)
            if "s_endpgm" in l:
                break
        counts, found = check_function(name, body, show_all)
        print("%-72s producers %s, %d hazard(s)" % (name[:72], " ".join("%s:%d" % kv for kv in sorted(counts.items())), len(found)))
        total += found
    for e in total[:60]:
        print("HAZARD", e)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
