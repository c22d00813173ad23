#!/bin/bash
# Box-side: same-box A/B of builds of libmetran_hip.so on c4_full_sym (all six outputs of configs[3]'s batch as packed-symmetric
# records: the split record filter + the RTS MFMA smoother), interleaved, then parity of the LAST library on the AOT shapes.
#   gpurun -- 'bash scripts/ab_c4f.sh ab/lib_A.so ab/lib_B.so ...'
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; log=gpurun_out/r06_ab_c4f.log; : > $log
run() {
  METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/$1 timeout 400 python bench.py --config c4f --packed-sym --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['models_per_s']), {k:round(v['ms'],2) for k,v in d['roofline']['kernels'].items()})" >> $log 2>&1
}
for i in 1 2; do for lib in "$@"; do run $lib; done; done
last="${@: -1}"
echo "== parity of $last (AOT shapes only)" >> $log
METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/$last timeout 1200 python -m pytest tests/test_hip_layouts.py tests/test_hip_parity.py tests/test_smoother_variants.py tests/test_gpu_property.py -q -m gpu -k "not 48 and not runtime and not jit" 2>&1 | tail -6 >> $log
cat $log
