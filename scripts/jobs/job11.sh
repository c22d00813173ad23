#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job11; mkdir -p $OUT
for v in v1 v2 v3; do
MK_WIDE_SMOOTHER=$v timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "c4_missing or n17" > $OUT/pytest_$v.log 2>&1; echo "$v rc=$?"; grep -E "passed|failed|Memory access|^E  " $OUT/pytest_$v.log | head -5
done
