#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job30; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2; do for cfg in c2 c3; do for sym in "" "--packed-sym"; do
  tag=${cfg}${sym:+_sym}
  timeout 300 python bench.py --config $cfg $sym --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$tag.json")); r=d["roofline"]
    print("$tag: ms/step %.3f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],3) for k,v in r["kernels"].items()}, "frac %.3f"%r["frac"])
except Exception as e:
    print("$tag: failed", e); print(open("$OUT/bench_$tag.err").read()[-800:])
PY
done; done; done
