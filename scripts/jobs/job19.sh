#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job19; mkdir -p $OUT
cd scripts/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/power_probe power_probe.hip 2> $GRAFT_REPO_ROOT/$OUT/build.err && timeout 120 /tmp/power_probe | tee $GRAFT_REPO_ROOT/$OUT/power_probe.log
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > $GRAFT_REPO_ROOT/$OUT/smi.log
