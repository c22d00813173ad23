#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_prof2; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | head -20
for v in mfma mfma16; do
  VARIANT=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/p_$v -- python scripts/jobs/r3_tune.py > $OUT/run_$v.log 2>&1
  python scripts/pmc_extract.py /tmp/p_$v $OUT/pmc_$v.json > /dev/null
  python - <<PY
import json
d=json.load(open("$OUT/pmc_$v.json"))
for k,v in d.items():
    if "smoother" in k or "filter" in k: print("$v", k[:70], {c:(round(x/1e6,1) if isinstance(x,float) and x>1e5 else x) for c,x in v.items()})
PY
done
tail -3 $OUT/run_mfma.log
