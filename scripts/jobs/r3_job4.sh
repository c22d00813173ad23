#!/bin/bash
# round 3, lease 4: wide adjoint (parity), calibration of 512 x (32,4) models: adjoint vs differenced gradients
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r3_job4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_adjoint.py tests/test_hip_solver.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest.log
for g in adjoint fd; do
  timeout 900 python scripts/bench_calibrate.py --batch 512 --series 32 --factors 4 --T 500 --missing 0.3 --gradient $g --fd-below 0 --maxiter 60 > $OUT/calib_wide_$g.json 2> $OUT/calib_wide_$g.err; echo "calib $g rc=$?"
  cat $OUT/calib_wide_$g.json; tail -3 $OUT/calib_wide_$g.err
done
