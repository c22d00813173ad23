#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job36; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2 3; do
timeout 120 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c4.json")); r=d["roofline"]
    print("c4: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()})
except Exception as e:
    print("bench failed", e); print(open("$OUT/bench_c4.err").read()[-500:])
PY
done
