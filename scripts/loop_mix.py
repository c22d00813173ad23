"""Dev helper: instruction mix of the largest loop of one kernel in a hipcc -save-temps .s file.
    python scripts/loop_mix.py file.s kernel_symbol_substring [top]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m]
for name in names:
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
    loop = body[best[1]:best[2]] if best else body
    c = collections.Counter()
    for l in loop:
        l = l.strip()
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(name)
    print("  loop instructions %d, VALU %d, MFMA %d, LDS %d, global %d, scratch %d" % (
        sum(c.values()), valu, sum(v for k, v in c.items() if k.startswith("v_mfma")),
        sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith("global_")),
        sum(v for k, v in c.items() if k.startswith("scratch_"))))
    print("  " + ", ".join("%s %d" % kv for kv in c.most_common(top)))
