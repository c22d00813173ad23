#!/bin/bash
# Box-side: rocprofv3 kernel-trace stats + HBM PMC passes of the default bench command; reduced
# summaries land in gpurun_out/ (copy them into profiles/<round>/ afterwards).
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof; mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $CMD > $OUT/bench_kt.log 2>&1
cp /tmp/kt/*/*kernel_stats.csv $OUT/kernel_stats_full.csv
head -1 $OUT/kernel_stats_full.csv > $OUT/kernel_stats.csv; grep "mk::" $OUT/kernel_stats_full.csv >> $OUT/kernel_stats.csv; rm $OUT/kernel_stats_full.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $CMD > /dev/null 2>&1
python scripts/pmc_extract.py /tmp/pf $OUT/pmc_fetch.json > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $CMD > /dev/null 2>&1
python scripts/pmc_extract.py /tmp/pw $OUT/pmc_write.json > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/ps -- $CMD > /dev/null 2>&1
python scripts/pmc_extract.py /tmp/ps $OUT/pmc_sq.json > /dev/null
cat $OUT/kernel_stats.csv; tail -1 $OUT/bench_kt.log | cut -c1-300
