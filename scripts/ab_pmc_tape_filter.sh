#!/bin/bash
# Box-side: SQ counters of the two tape writers (bench.py --tape-filter state | observable) on configs[3], two separate PMC passes each
# (counters only, no trace domain), into gpurun_out/pmc_tape_filter_<variant>_{a,b}.json.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
CFG=${1:-c4}
for v in state observable; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --config $CFG --tape-filter $v --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-live-traffic"
  cd /tmp; rm -rf /tmp/ps1 /tmp/ps2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/ps1 -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d /tmp/ps2 -- $CMD > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/pmc_extract.py /tmp/ps1 gpurun_out/pmc_tape_filter_${CFG}_${v}_a.json > /dev/null
  python scripts/pmc_extract.py /tmp/ps2 gpurun_out/pmc_tape_filter_${CFG}_${v}_b.json > /dev/null
  python - <<PY
import json
for t in "ab":
    d=json.load(open("gpurun_out/pmc_tape_filter_${CFG}_${v}_%s.json"%t))
    for k,c in d.items():
        if "filter" in k: print("$v", k[:60], {a:round(b/1e6,2) for a,b in c.items() if isinstance(b,(int,float))})
PY
done
