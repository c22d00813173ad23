#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job16; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for cfg in c2 c3; do for sym in "" "--packed-sym"; do for mode in duo record; do
  tag=${cfg}_${mode}${sym:+_sym}
  MK_SMOOTHER16=$mode timeout 300 python bench.py --config $cfg $sym --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$tag.json")); r=d["roofline"]
    print("$tag: ms/step %.3f models/s %.0f"%(d["ms_per_step"], d["value"]), {k:round(v["ms"],3) for k,v in r["kernels"].items()}, "frac", r["frac"])
except Exception as e:
    print("$tag: failed", e); print(open("$OUT/bench_$tag.err").read()[-800:])
PY
done; done; done
