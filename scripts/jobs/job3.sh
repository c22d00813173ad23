#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/job3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_singular.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 600 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --config c4 --T 200 --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $CMD > $GRAFT_REPO_ROOT/$OUT/kt.log 2>&1
cp /tmp/kt/*/*kernel_stats.csv $GRAFT_REPO_ROOT/$OUT/kernel_stats_T200.csv 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d /tmp/p1 -- $CMD > $GRAFT_REPO_ROOT/$OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/p2 -- $CMD > $GRAFT_REPO_ROOT/$OUT/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_extract.py /tmp/p1 $OUT/pmc_a.json > /dev/null 2>&1
python scripts/pmc_extract.py /tmp/p2 $OUT/pmc_b.json > /dev/null 2>&1
tail -8 $OUT/pytest.log
python - <<PY
import json
d=json.load(open("$OUT/bench_c4.json")); r=d["roofline"]
print("c4: ms/step %.2f models/s %.0f"%(d["ms_per_step"], d["models_per_s"]), {k:round(v["ms"],2) for k,v in r["kernels"].items()})
for f in ("pmc_a","pmc_b"):
    try:
        x=json.load(open("$OUT/%s.json"%f))
        for k,v in x.items(): print(k[:70]); print("   ",{c:round(y,1) for c,y in v.items()})
    except Exception as e: print(f,e)
PY
grep mk:: $OUT/kernel_stats_T200.csv | cut -c1-200; tail -3 $OUT/p2.log
