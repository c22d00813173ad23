#!/bin/bash
# A timing build of libmetran_hip.so: mk_split.hip recompiled with -DMK_TUNE=<mask> (mk_internal.h; wrong results, timing only),
# linked with the objects of the last `make`.   bash scripts/build_tune_lib.sh <mask> <out.so> [file.hip]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); T=$1; OUT=$2; SRC=${3:-mk_split}
B=/tmp/tune_$T; mkdir -p $B
[ -f $B/$SRC.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -I$ROOT/metran_amd/csrc -Wall -Wno-unused-parameter \
  -DMK_TUNE=$T -c $ROOT/metran_amd/csrc/$SRC.hip -o $B/$SRC.o
OBJS=$(ls $ROOT/build/csrc/*.o $ROOT/build/csrc/wide_p*/mk_wide.o | grep -v "/$SRC.o" | grep -v "amdgcn")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $B/$SRC.o $OBJS
