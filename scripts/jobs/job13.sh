#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job13
timeout 600 python scripts/probe_wide_occupancy.py 2>&1 | tee gpurun_out/job13/occ.log
