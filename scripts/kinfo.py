"""Dev helper: per-kernel resource usage (VGPRs, scratch, LDS, occupancy, code size) from hipcc's -save-temps assembly."""
import re, sys
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_ZN2mk\w+):", line)
    if m:
        cur = m.group(1)
    m = re.match(r"^; (codeLenInByte|NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs)\D*(\d+)", line)
    if m and cur:
        rows.setdefault(cur, {})[m.group(1)] = int(m.group(2))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in rows.items():
    if pat in k:
        print(k[:90], v)
