#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job20; mkdir -p $OUT
cd scripts/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma44 mfma44.hip 2> $GRAFT_REPO_ROOT/$OUT/build.err && timeout 120 /tmp/mfma44 | tee $GRAFT_REPO_ROOT/$OUT/mfma44.log
