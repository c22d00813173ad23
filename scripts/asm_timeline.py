"""Dev helper: coarse timeline of instruction categories of one kernel in hipcc's assembly (windows of W instructions).
   python scripts/asm_timeline.py file.s kernel_substring [W]"""
import re, sys
S, pat = sys.argv[1], sys.argv[2]
W = int(sys.argv[3]) if len(sys.argv) > 3 else 500
lines = open(S).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN2mk\w+:", l) and pat in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith("\t.section") and i > start + 10][0]
cats = (("scratch", "scratch_"), ("mfma", "v_mfma"), ("ds_rd", "ds_read"), ("ds_wr", "ds_write"), ("rdlane", "v_readlane"), ("fma", "v_fma"),
        ("glob", "global_"), ("nop", "s_nop"), ("wait", "s_waitcnt"))
cur = dict.fromkeys([c for c, _ in cats], 0); n = 0
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    n += 1
    for c, key in cats:
        if key in t:
            cur[c] += 1
    if n % W == 0:
        print(n, cur); cur = dict.fromkeys(cur, 0)
print(n, cur)
