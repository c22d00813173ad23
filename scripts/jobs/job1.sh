#!/bin/bash
# Box-side job: ubench + full GPU test suite + one bench line per configuration.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job1; mkdir -p $OUT
./scripts/ubench/mfma_f64 > $OUT/ubench_mfma.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 300 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 300 python bench.py --config c3 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
timeout 600 python bench.py --config c4 --steps 2 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
timeout 300 python bench.py --config c5 --steps 2 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
tail -25 $OUT/ubench_mfma.log; tail -15 $OUT/pytest.log
for c in c2 c3 c4 c5; do echo "== $c"; tail -c 1500 $OUT/bench_$c.json; tail -3 $OUT/bench_$c.err; done
