#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job29
timeout 300 python scripts/probe_ramp.py 2>&1 | grep launches | tee gpurun_out/job29/ramp.log
