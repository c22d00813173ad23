#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job38
timeout 300 python scripts/probe_calibrate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/job38/cal.log | awk 'NR%5==1 || /wall/'
