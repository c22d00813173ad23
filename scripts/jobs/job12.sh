#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job12; mkdir -p $OUT
for p in off 0 1 2 3 5 8; do
  if [ $p = off ]; then unset MK_WIDE_PRIO; else export MK_WIDE_PRIO=$p; fi
  timeout 300 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$p.json 2> $OUT/bench_$p.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$p.json")); r=d["roofline"]
print("prio $p: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()})
PY
done
