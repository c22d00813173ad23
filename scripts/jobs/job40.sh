#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job40; mkdir -p $OUT
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; echo "rc=$?"
tail -c 300 $OUT/bench_torchrun1.json; echo; grep -o '"rccl_ranks": [0-9]*' $OUT/bench_torchrun1.json; tail -3 $OUT/bench_torchrun1.err
timeout 300 python -m pytest tests/test_hip_layouts.py -m gpu -q -k "timing" 2>&1 | tail -2
