#!/bin/bash
# Dev helper: compile the kernels of ONE (N,K) shape to assembly and print resource usage.
#   scripts/compile_shape.sh N K [wide|all] [extra hipcc flags...]      (wide: mk_wide.hip only)
set -e
N=$1; K=$2; WHAT=${3:-all}; shift 3 || shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/mkshape_${N}_${K}_${WHAT}; mkdir -p $OUT
cd $OUT
if [ "$WHAT" = wide ]; then SRC=mk_wide; DEF=""; else SRC=mk_kernels; DEF="-DMK_SHAPE_MODULE"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -I$ROOT/metran_amd/csrc \
  $DEF "-DMK_SHAPES(X)=X($N,$K)" -save-temps=obj -Wno-unused-command-line-argument "$@" -c $ROOT/metran_amd/csrc/$SRC.hip -o $OUT/mod.o
python3 $ROOT/scripts/kinfo.py $OUT/$SRC-hip-amdgcn-amd-amdhsa-gfx950.s
python3 $ROOT/scripts/check_dpp_hazards.py $OUT/$SRC-hip-amdgcn-amd-amdhsa-gfx950.s | grep -v " 0 hazard" || true
