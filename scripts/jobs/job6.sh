#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_layouts.py -m gpu -q -x > $OUT/pytest_layouts.log 2>&1; echo "rc=$?" >> $OUT/pytest_layouts.log
timeout 900 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 300 python bench.py --no-cpu-baseline --packed-sym > $OUT/bench_c2_sym.json 2> $OUT/bench_c2_sym.err
timeout 300 python bench.py --config c3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 300 python bench.py --config c3 --no-cpu-baseline --packed-sym > $OUT/bench_c3_sym.json 2> $OUT/bench_c3_sym.err
grep -E "^E  |passed|failed|FAILED|rc=" $OUT/pytest_layouts.log | head -30
grep -E "passed|failed|FAILED|rc=" $OUT/pytest.log | head -20
for c in c2 c2_sym c3 c3_sym; do python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$c.json")); r=d["roofline"]
    print("$c: ms/step %.3f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:(round(v["ms"],3), round(v["GBps"])) for k,v in r["kernels"].items()}, "frac %.3f"%r["frac"])
except Exception as e: print("$c", e); print(open("$OUT/bench_$c.err").read()[-500:])
PY
done
