#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job2; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
MK_WIDE_SMOOTHER=v1 timeout 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c4_v1.json 2> $OUT/bench_c4_v1.err
timeout 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
tail -12 $OUT/pytest.log
for c in c4_v1 c4; do echo "== $c"; python - <<PY
import json
d=json.load(open("$OUT/bench_$c.json")); r=d["roofline"]
print("ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()}, "8d frac %.3f"%r["survey_8d_full_output_accounting"]["frac_of_peak"])
PY
tail -2 $OUT/bench_$c.err; done
