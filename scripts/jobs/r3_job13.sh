#!/bin/bash
# round 3, lease 13: projection constants in LDS (wide smoother): A/B, live HBM traffic, parity of the projection paths
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --config c4 --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['models_per_s']), {k:round(v['ms'],2) for k,v in d['roofline']['kernels'].items()})"
}
METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/libmetran_hip_base.so run base
run new
timeout 600 python bench.py --config c4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('live', round(d['models_per_s']), r.get('traffic'), r.get('algorithmic_bytes'), {k:v for k,v in r.items() if 'traffic' in k})"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_smoother_variants.py tests/test_factoranalysis_gpu.py -m gpu -q -x 2>&1 | tail -3
