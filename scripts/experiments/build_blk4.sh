#!/bin/bash
# Builds ab/libmetran_hip_blk4.so: the library with the EXPERIMENTAL 4x4x4 MFMA block path of the wide smoother compiled in
# (-DMK_EXPERIMENTAL_BLK4: mk_set_kernel_variant(ctx, MK_VARIANT_WIDE_SMOOTHER, 3 | 4) = block path with | without the lane
# fold).  ab/ is git-ignored and travels to the GPU box; the product library (metran_amd/libmetran_hip.so) is untouched.
#   bash scripts/experiments/build_blk4.sh && gpurun -- 'bash scripts/experiments/ab_blk4.sh'
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p $ROOT/ab
make -C $ROOT/metran_amd/csrc BUILD=$ROOT/build/csrc_blk4 OUT=$ROOT/ab/libmetran_hip_blk4.so EXTRA=-DMK_EXPERIMENTAL_BLK4 -j4
ls -la $ROOT/ab/libmetran_hip_blk4.so
