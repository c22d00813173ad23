#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job25
timeout 300 python scripts/probe_out3.py 2>&1 | tee gpurun_out/job25/out3.log
