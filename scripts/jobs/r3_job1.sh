#!/bin/bash
# round 3, lease 1: the whole GPU tier (new: multi-factor parity, variant API, RCCL world-1, bench floors) + the default bench line
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r3_job1; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("c2", d["models_per_s"], d["roofline"]["frac"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
for k,v in d.get("secondary",{}).items():
    print(k, v.get("models_per_s"), v.get("filter_ms"), v.get("smoother_ms"), v.get("roofline",{}).get("frac"), v.get("error"))
print("cpu", d.get("cpu_baseline",{}).get("models_per_s"), d.get("loglik_max_rel_err"))
PY
