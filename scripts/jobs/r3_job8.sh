#!/bin/bash
# round 3, lease 8: wide smoother back substitution in dot form over L^T in LDS -- parity and timing
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python scripts/experiments/wide_phase_timing.py 2>&1 | tail -1
timeout 300 python bench.py --config c4 --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4', round(d['models_per_s']), {k:round(v['ms'],2) for k,v in d['roofline']['kernels'].items()})"
timeout 900 python -m pytest tests/test_smoother_variants.py tests/test_hip_parity.py tests/test_hip_singular.py tests/test_hip_layouts.py -m gpu -q -x 2>&1 | tail -3
