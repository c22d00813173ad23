#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_factoranalysis_gpu.py tests/test_batch_facade.py tests/test_reference_dropin_gpu.py -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 300 python scripts/bench_calibrate.py --batch 8192 > $OUT/calib.json 2> $OUT/calib.err
grep -E "^E  |passed|failed|Error|FAILED|rc=" $OUT/pytest.log | head -60
tail -c 800 $OUT/calib.json; tail -3 $OUT/calib.err
