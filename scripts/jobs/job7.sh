#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_layouts.py -m gpu -q > $OUT/pytest_layouts.log 2>&1; echo "rc=$?" >> $OUT/pytest_layouts.log
grep -E "^E  |passed|failed|FAILED|rc=" $OUT/pytest_layouts.log | head -30
