#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job37; mkdir -p $OUT
# smallest wide case first, each under its own short timeout
timeout 120 python -m pytest tests/test_hip_layouts.py -m gpu -q -x -k "32 or 14" > $OUT/pytest1.log 2>&1; echo "rc=$?" >> $OUT/pytest1.log; tail -3 $OUT/pytest1.log
timeout 180 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "c4 or n17" > $OUT/pytest2.log 2>&1; echo "rc=$?" >> $OUT/pytest2.log; tail -3 $OUT/pytest2.log
timeout 120 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c4.json")); r=d["roofline"]
    print("c4: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()})
except Exception as e:
    print("bench failed", e); print(open("$OUT/bench_c4.err").read()[-500:])
PY
