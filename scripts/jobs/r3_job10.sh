#!/bin/bash
# round 3, lease 10: same-box A/B of the wide smoother (committed library vs the working tree's), interleaved twice
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --config c4 --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['models_per_s']), {k:round(v['ms'],2) for k,v in d['roofline']['kernels'].items()})"
}
for i in 1 2; do
  METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/libmetran_hip_base.so run base
  run new
done
