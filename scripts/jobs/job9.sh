#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job9; mkdir -p $OUT
for lib in base w3 w4; do
  if [ $lib = base ]; then unset METRAN_HIP_LIBRARY; else export METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/metran_amd/libmetran_hip_$lib.so; fi
  timeout 600 python bench.py --config c4 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_c4_$lib.json 2> $OUT/bench_c4_$lib.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c4_$lib.json")); r=d["roofline"]
    print("$lib", "ms/step %.2f"%d["ms_per_step"], {k:round(v["ms"],2) for k,v in r["kernels"].items()})
except Exception as e: print("$lib", e)
PY
done
