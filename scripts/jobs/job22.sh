#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job22; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for cfg in c2 c3 c4 c5; do
  st=10; [ $cfg = c4 ] && st=2; [ $cfg = c5 ] && st=2
  timeout 300 python bench.py --config $cfg --steps $st --warmup 2 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$cfg.json")); r=d["roofline"]
    print("$cfg: ms/step %.3f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],3) for k,v in r["kernels"].items()}, "frac %.3f"%r["frac"])
except Exception as e:
    print("$cfg: failed", e); print(open("$OUT/bench_$cfg.err").read()[-800:])
PY
done
