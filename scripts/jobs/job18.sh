#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job18; mkdir -p $OUT
cd scripts/ubench
for f in issue_rate dpp_rate; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$f $f.hip 2> $GRAFT_REPO_ROOT/$OUT/$f.err && timeout 120 /tmp/$f | tee $GRAFT_REPO_ROOT/$OUT/$f.log
done
