"""Resource usage (VGPRs, spills, scratch, LDS) of the kernels in a hipcc -save-temps assembly file, from its metadata:
python scripts/kres.py build/csrc/mk_dk-hip-amdgcn-amd-amdhsa-gfx950.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
md = s[s.find("amdhsa.kernels"):]
for blk in md.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt not in name:
        continue
    g = lambda k: re.search(k + r":\s*(\d+)", blk)  # noqa: E731
    keys = ["\\.vgpr_count", "\\.sgpr_count", "\\.private_segment_fixed_size", "\\.group_segment_fixed_size", "\\.vgpr_spill_count"]
    print(name[:70], {k.strip("\\."): int(g(k).group(1)) for k in keys if g(k)})
