#!/bin/bash
# Box-side: rocprofv3 kernel-trace stats of the non-headline measurements quoted in DESIGN.md section 6
# (state outputs of configs[3]'s batch, adjoint gradient, the drop-in on examples/data, batched calibration).
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_extra; mkdir -p $OUT
run() { # name, command...
    local name=$1; shift
    rm -rf /tmp/kx; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kx -- "$@" > $OUT/$name.log 2>&1
    local f=$(ls /tmp/kx/*/*kernel_stats.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then head -1 $f > $OUT/${name}_kernel_stats.csv; grep "mk::" $f >> $OUT/${name}_kernel_stats.csv; fi
    tail -2 $OUT/$name.log | cut -c1-400
}
run c4_state_variances python bench.py --config c4s --no-cpu-baseline --no-secondary --no-live-traffic --steps 3 --warmup 1
run adjoint_gradient python scripts/probe.py kernels --what grad --batch 8192
run dropin_examples_data python scripts/probe.py dropin
run calibrate_batch python scripts/bench_calibrate.py
cat $OUT/*_kernel_stats.csv | cut -c1-170
