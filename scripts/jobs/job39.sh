#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job39; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_solver.py tests/test_batch_facade.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 300 python scripts/bench_calibrate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cal.json
