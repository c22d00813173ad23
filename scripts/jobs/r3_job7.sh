#!/bin/bash
# round 3, lease 7: the whole GPU tier on the final tree + the default bench line (live PMC traffic, secondary configurations)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r3_job7; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -16 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
r=d["roofline"]
print("c2", round(d["models_per_s"]), "frac", round(r["frac"],3), "traffic", r["traffic"], r["kernels"][r["kernel"]].get("traffic_source","")[:60], r["kernels"][r["kernel"]].get("live_traffic_note"))
for k,v in d.get("secondary",{}).items():
    print(k, v.get("models_per_s"), v.get("filter_ms"), v.get("smoother_ms"), v.get("roofline",{}).get("frac"), v.get("error"))
PY
tail -3 $OUT/bench_default.err
