#!/bin/bash
# Box-side: SQ counters of the (32,4) kernels (one model per wavefront), T=200 so that all outputs fit.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_c4; mkdir -p $OUT
export B=4096 T=200
CMD="python scripts/probe_c4.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/p1 -- $CMD > $OUT/run1.log 2>&1
python scripts/pmc_extract.py /tmp/p1 $OUT/pmc_sq_a.json > /dev/null
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/p2 -- $CMD > $OUT/run2.log 2>&1
python scripts/pmc_extract.py /tmp/p2 $OUT/pmc_sq_b.json > /dev/null
python - <<'PY'
import json
for f in ("gpurun_out/prof_c4/pmc_sq_a.json", "gpurun_out/prof_c4/pmc_sq_b.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, e); continue
    for k, v in d.items():
        print(k[:60])
        print("   ", {c: (round(x, 1) if isinstance(x, float) else x) for c, x in v.items()})
PY
tail -3 $OUT/run1.log
