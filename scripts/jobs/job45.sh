#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/job45
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -20
for lib in head new head new; do
  export METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/metran_amd/libmetran_hip_$lib.so
  for cfg in c2 c5; do
    timeout 300 python bench.py --config $cfg --no-cpu-baseline > gpurun_out/job45/b.json 2> gpurun_out/job45/b.err
    python - <<PY
import json
d=json.load(open("gpurun_out/job45/b.json")); r=d["roofline"]
print("$lib $cfg: ms/step %.3f"%d["ms_per_step"], {k:round(v["ms"],3) for k,v in r["kernels"].items()})
PY
  done
done | tee gpurun_out/job45/ab.log
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^=\|^$" | head -20
