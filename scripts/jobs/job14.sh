#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job14
for t in 0 1 2 4 6 7; do
  echo "== MK_WIDE_TUNE=$t"
  MK_WIDE_TUNE=$t BS=1024,1792 timeout 300 python scripts/probe_wide_occupancy.py 2>&1 | grep "^B"
done | tee gpurun_out/job14/tune.log
