#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job15; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_layouts.py -m gpu -q -x -k "c4 or n17 or golden_synthetic or wide or 32 or sym or var" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
BS=1024,1792,2048,4096 timeout 300 python scripts/probe_wide_occupancy.py 2>&1 | grep "^B" | tee $OUT/occ.log
timeout 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python - <<PY
import json
d=json.load(open("$OUT/bench_c4.json")); r=d["roofline"]
print("c4: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()})
PY
