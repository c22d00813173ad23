#!/bin/bash
# Box-side: same-box A/B of two builds of libmetran_hip.so on one bench configuration, interleaved (leases differ by up to
# 25 % on compute-bound kernels, DESIGN.md section 6: only a comparison on one box means anything).
#   gpurun -- 'bash scripts/ab_bench.sh c4 ab/libmetran_hip_base.so [pytest files...]'
# BASE: a library built from the commit to compare against (make -C metran_amd/csrc BUILD=/tmp/x OUT=$PWD/ab/libmetran_hip_base.so
# in a stash / worktree of that commit); the working tree's library is the other side.  Optional parity tests afterwards.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
CFG=${1:-c4}; BASE=${2:-ab/libmetran_hip_base.so}; shift 2 || true
run() {
  timeout 300 python bench.py --config $CFG --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['models_per_s']), {k:round(v['ms'],2) for k,v in d['roofline']['kernels'].items()})"
}
for i in 1 2; do
  METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/$BASE run base
  run new
done
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -m gpu -q -x 2>&1 | tail -5; fi
