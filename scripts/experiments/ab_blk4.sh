#!/bin/bash
# Box-side (round 4, first call): the experimental block-path library against the product kernels.
#   1. determinism + equality at FULL occupancy (the round-3 failure: 4096 models, two wavefronts per SIMD), three runs
#   2. parity against the oracle on the GPU test's own seeded case
#   3. same-box A/B timing at configs[3] size
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export METRAN_HIP_LIBRARY=$GRAFT_REPO_ROOT/ab/libmetran_hip_blk4.so
mkdir -p gpurun_out
timeout 900 python scripts/experiments/blk4_determinism_check.py 2>&1 | tee gpurun_out/blk4_determinism.log
timeout 900 python scripts/experiments/ab_wide_variants.py ${1:-2000} mfma mfma_blk4 mfma_blk4_unfolded 2>&1 | tee gpurun_out/blk4_ab.log
timeout 900 python -m pytest tests/test_smoother_variants.py -m gpu -q -x -k "block_path" 2>&1 | tail -5 | tee gpurun_out/blk4_pytest.log
