#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job42
for fd in 0 4096 16384; do echo "== FD_BELOW=$fd"; FD_BELOW=$fd timeout 300 python scripts/probe_calibrate.py 2>&1 | grep -v amdgpu.ids | grep "wall\|frac"; done | tee gpurun_out/job42/hybrid.log
