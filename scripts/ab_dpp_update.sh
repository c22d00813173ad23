#!/bin/bash
# Box-side: same-box A/B of builds of libmetran_hip.so on the tape paths of configs[3] (projection and state outputs), then the tape
# parity tests under the LAST library.  Every experiment of round 6's last sessions ran through this (profiles/r06/ab_*.log).
#   gpurun -- 'bash scripts/ab_dpp_update.sh <name> ab/lib_BASE.so ab/lib_X.so [...]'
name=${1:-ab}; shift
mkdir -p gpurun_out
log=gpurun_out/r06_ab_${name}.log
: > $log
python scripts/ab_libs.py "$@" >> $log 2>&1
python scripts/ab_libs.py "$@" --state >> $log 2>&1
last="${@: -1}"
echo "== tape parity of $last" >> $log
METRAN_HIP_LIBRARY=$PWD/$last timeout 900 python -m pytest tests/test_dk_tape.py tests/test_gpu_property.py -q -m gpu -k "not 48 and not 19 and not 20 and not 17 and (dk_tape or 32x4 or 14x3)" 2>&1 | tail -5 >> $log
cat $log
