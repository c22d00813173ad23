#!/bin/bash
# round 3, lease 2: split-layout wide filter -- parity (whole GPU tier) + c4 bench A/B against the lane-per-state kernel
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r3_job2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_gpu.py --deselect tests/test_rccl_gpu.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest.log
timeout 600 python bench.py --config c4 --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c4.json")); r=d["roofline"]
    print("c4 split", "models/s %.0f"%d["models_per_s"], {k:round(v["ms"],2) for k,v in r["kernels"].items()})
except Exception as e: print("c4", e); print(open("$OUT/bench_c4.err").read()[-2000:])
PY
