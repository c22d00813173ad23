#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash scripts/collect_profiles.sh c2 c2sym c3 c3sym c4 c5 > gpurun_out/prof_r02_collect.log 2>&1
tail -40 gpurun_out/prof_r02_collect.log
