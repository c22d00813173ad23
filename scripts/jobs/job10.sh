#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_layouts.py tests/test_hip_singular.py -m gpu -q -x -k "c4 or n17 or golden_synthetic or 32 or 14 or runtime or heywood or degenerate" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
for v in v2 v3; do
MK_WIDE_SMOOTHER=$v timeout 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c4_$v.json 2> $OUT/bench_c4_$v.err
python - <<PY
import json
d=json.load(open("$OUT/bench_c4_$v.json")); r=d["roofline"]
print("$v c4: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(x["ms"],2) for k,x in r["kernels"].items()})
PY
done
grep -E "^E  |passed|failed|FAILED|rc=" $OUT/pytest.log | head -20
