#!/bin/bash
# round 3, lease 5: tiled sweeps for 11 <= n <= 16 (parity of run-time shapes, A/B timing), wide adjoint test fix
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r3_job5; mkdir -p $OUT
export METRAN_HIP_CACHE=/tmp/mkjit
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_adjoint.py tests/test_smoother_variants.py -m gpu -q -x -k "runtime or adjoint or variant" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest.log
timeout 900 python scripts/probe_sweeps.py > $OUT/sweeps_tiled.jsonl 2> $OUT/sweeps_tiled.err; echo "tiled rc=$?"; cat $OUT/sweeps_tiled.jsonl
METRAN_HIP_JIT_FLAGS=-DMK_NO_TILED_SWEEPS timeout 900 python scripts/probe_sweeps.py > $OUT/sweeps_per_fma.jsonl 2> $OUT/sweeps_per_fma.err; echo "per-fma rc=$?"; cat $OUT/sweeps_per_fma.jsonl
tail -3 $OUT/sweeps_per_fma.err
