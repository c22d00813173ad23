#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job43
for rep in 1 2; do for lib in base attr; do
  if [ $lib = attr ]; then export METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/metran_amd/libmetran_hip_attr.so; else unset METRAN_HIP_LIBRARY; fi
  echo "== $lib"; timeout 300 python scripts/probe_out3.py 2>&1 | grep "^B "
done; done | tee gpurun_out/job43/attr.log
