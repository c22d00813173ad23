"""Dev helper: every loop (backward branch) of one kernel in a hipcc -save-temps .s file with its instruction mix.
    python scripts/loops_all.py file.s kernel_symbol_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
for name in [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m]:
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    print(name)
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loop = body[labels[m.group(1)]:k + 1]
            c = collections.Counter()
            for x in loop:
                x = x.split(";")[0].strip()
                if not x or x.startswith(".") or x.endswith(":"):
                    continue
                c[x.split()[0]] += 1
            valu = sum(v for kk, v in c.items() if kk.startswith("v_") and not kk.startswith("v_mfma"))
            dpp = sum(v for kk, v in c.items() if "f64_dpp" in kk)
            print("  loop %s lines %d-%d: %d instr, VALU %d (dpp64 %d), SALU %d, LDS %d, VMEM %d, s_nop %d, waitcnt %d" % (
                m.group(1), labels[m.group(1)], k, sum(c.values()), valu, dpp,
                sum(v for kk, v in c.items() if kk.startswith("s_") and kk not in ("s_nop", "s_waitcnt")),
                sum(v for kk, v in c.items() if kk.startswith("ds_")),
                sum(v for kk, v in c.items() if kk.startswith(("global_", "buffer_"))), c["s_nop"], c["s_waitcnt"]))
            if len(sys.argv) > 3:
                print("     " + ", ".join("%s %d" % kv for kv in c.most_common(30)))
