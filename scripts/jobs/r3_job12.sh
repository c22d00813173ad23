#!/bin/bash
# round 3, lease 12: parallel-ordering Jacobi in the factor analysis -- parity and throughput; wide packed-symmetric parity
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_factoranalysis_gpu.py tests/test_hip_layouts.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python scripts/probe_factor.py 2>&1 | tail -4
