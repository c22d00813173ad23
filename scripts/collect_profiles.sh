#!/bin/bash
# Box-side: the round's profiles.  For every bench configuration: rocprofv3 kernel-trace stats, HBM traffic from two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with trace domains), the bench line of the same
# command; SQ counters for c2 and c4.  Everything lands in gpurun_out/prof_${ROUND}/ -- copy into profiles/${ROUND}/.
#   ROUND=r06 bash scripts/collect_profiles.sh [configs...]        (default: c2 c3 c4 c4s c5 c4fsym; c2sym / c3sym / c4fsym: packed-symmetric records)
set -u
ROUND=${ROUND:-r06}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${ROUND}; mkdir -p $OUT
CFGS="${@:-c2 c3 c4 c4s c5 c4fsym}"
for tag in $CFGS; do
  cfg=${tag%sym}; symflag=""; sym=0
  if [ "$tag" != "$cfg" ]; then symflag="--packed-sym"; sym=1; fi
  # the default bench.py run of each configuration (c2/c3: 50 timed + 10 warm-up steps -- the GPU needs ~10 launches after
  # idling to reach its steady clocks, `scripts/probe.py kernels --ramp 40`; the kernel-trace averages include those warm-up launches)
  steps=50; warm=10; [ $cfg = c4 ] && steps=3 && warm=1; [ $cfg = c4s ] && steps=3 && warm=1; [ $cfg = c5 ] && steps=3 && warm=1; [ $cfg = c4f ] && steps=3 && warm=1
  CMD="python $GRAFT_REPO_ROOT/bench.py --config $cfg $symflag --steps $steps --warmup $warm --no-cpu-baseline --no-secondary --no-live-traffic"
  cd /tmp
  rm -rf /tmp/kt /tmp/pf /tmp/pw
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $CMD > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/${tag}_kt.err
  f=$(ls /tmp/kt/*/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/${tag}_kernel_stats.csv; grep "mk::" $f >> $OUT/${tag}_kernel_stats.csv; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $CMD > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/pmc_extract.py /tmp/pf /tmp/${tag}_fetch.json > /dev/null
  python scripts/pmc_extract.py /tmp/pw /tmp/${tag}_write.json > /dev/null
  python scripts/merge_pmc.py $cfg $sym /tmp/${tag}_fetch.json /tmp/${tag}_write.json $OUT/pmc_hbm_${tag}.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (two separate passes) -- $CMD"
  if [ $cfg = c2 ] || [ $cfg = c4 ] || [ $cfg = c4s ] || [ $cfg = c4f ]; then
    cd /tmp; rm -rf /tmp/ps1 /tmp/ps2
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/ps1 -- $CMD > /dev/null 2>&1
    rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d /tmp/ps2 -- $CMD > /dev/null 2>&1
    cd $GRAFT_REPO_ROOT
    python scripts/pmc_extract.py /tmp/ps1 $OUT/pmc_sq_${tag}_a.json > /dev/null
    python scripts/pmc_extract.py /tmp/ps2 $OUT/pmc_sq_${tag}_b.json > /dev/null
  fi
  # the bench line of the plain command (with the CPU baselines for the headline configuration), reading the traffic just taken
  mkdir -p profiles/${ROUND}; cp $OUT/pmc_hbm_${tag}.json profiles/${ROUND}/
  extra="--no-cpu-baseline"; [ $tag = c2 ] && extra=""
  python bench.py --config $cfg $symflag $extra --no-secondary > $OUT/bench_${tag}.json 2> $OUT/bench_${tag}.err
  echo "== $tag"; cat $OUT/${tag}_kernel_stats.csv | cut -c1-150; tail -c 400 $OUT/bench_${tag}.json | head -c 400; echo
done
# the default command as the driver runs it (headline + the secondary configurations), under kernel-trace and plain
cd /tmp; rm -rf /tmp/ktd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktd -- python $GRAFT_REPO_ROOT/bench.py --no-live-traffic > $OUT/default_bench_under_rocprof.json 2> $OUT/default_kt.err
f=$(ls /tmp/ktd/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -1 $f > $OUT/default_kernel_stats.csv; grep "mk::" $f >> $OUT/default_kernel_stats.csv; fi
cd $GRAFT_REPO_ROOT
python bench.py --transfers > $OUT/bench_default.json 2> $OUT/bench_default.err   # (--transfers: adds the pcie object)
echo "== default"; cat $OUT/default_kernel_stats.csv | cut -c1-160
