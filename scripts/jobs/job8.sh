#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job8; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
for rep in 1 2; do
for lib in prev new; do
  if [ $lib = prev ]; then export METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/metran_amd/libmetran_hip_prev.so; else unset METRAN_HIP_LIBRARY; fi
  for c in c2 c3; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --steps 20 > $OUT/bench_${c}_${lib}_$rep.json 2> $OUT/bench_${c}_${lib}_$rep.err
  done
done
done
unset METRAN_HIP_LIBRARY
grep -E "^E  |passed|failed|FAILED|rc=" $OUT/pytest.log | head -20
for f in $OUT/bench_*.json; do python - <<PY
import json,os
try:
    d=json.load(open("$f")); r=d["roofline"]
    print(os.path.basename("$f"), "ms/step %.3f"%d["ms_per_step"], {k:round(v["ms"],3) for k,v in r["kernels"].items()})
except Exception as e: print("$f", e)
PY
done
