#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job32; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_layouts.py tests/test_hip_singular.py -m gpu -q -x -k "c4 or n17 or golden_synthetic or wide or 32 or sym or var or singular or heywood" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2; do
timeout 600 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python - <<PY
import json
d=json.load(open("$OUT/bench_c4.json")); r=d["roofline"]
print("c4: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()})
PY
done
