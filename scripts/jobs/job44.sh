#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job44
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/job44/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/job44/pytest.log
tail -4 gpurun_out/job44/pytest.log
timeout 300 python scripts/probe_out3.py 2>&1 | grep "^B " | tee gpurun_out/job44/out3.log
bash scripts/collect_profiles.sh c2 c2sym c3 c3sym c4 c5 > gpurun_out/prof_r02_collect.log 2>&1
grep -c "^== " gpurun_out/prof_r02_collect.log
