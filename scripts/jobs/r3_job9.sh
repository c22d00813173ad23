#!/bin/bash
# round 3, lease 9: wide smoother structured products (series tiles on the matrix cores, factor border on the vector pipe)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 python bench.py --config c4 --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4', round(d['models_per_s']), {k:round(v['ms'],2) for k,v in d['roofline']['kernels'].items()})"
timeout 1500 python -m pytest tests/test_smoother_variants.py tests/test_hip_parity.py tests/test_hip_singular.py tests/test_hip_layouts.py -m gpu -q -x 2>&1 | tail -5
