#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job31; mkdir -p $OUT
for rep in 1 2 3; do for lib in base unroll; do for cfg in c2 c3; do
  if [ $lib = unroll ]; then export METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/metran_amd/libmetran_hip_unroll.so; else unset METRAN_HIP_LIBRARY; fi
  timeout 300 python bench.py --config $cfg --no-cpu-baseline > $OUT/bench_${cfg}_$lib.json 2> $OUT/bench_${cfg}_$lib.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${cfg}_$lib.json")); r=d["roofline"]
    print("$cfg $lib: ms/step %.3f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],3) for k,v in r["kernels"].items()})
except Exception as e:
    print("$cfg $lib: failed", e); print(open("$OUT/bench_${cfg}_$lib.err").read()[-800:])
PY
done; done; done
