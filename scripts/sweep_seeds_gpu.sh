mkdir -p gpurun_out
log=gpurun_out/r06_sweep_seeds_final_tree_2.log
: > $log
for s in $(seq 161 200); do
  echo "== seed $s" >> $log
  METRAN_SWEEP_SEED=$s timeout 300 python -m pytest tests/test_gpu_property.py -q 2>&1 | grep -E "^E  |FAILED|passed|failed|Error" | head -20 >> $log
done
tail -80 $log
