mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_tier.log 2>&1; tail -5 gpurun_out/r06_gpu_tier.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
ROUND=r06 bash scripts/collect_profiles.sh > gpurun_out/collect_final_tree.log 2>&1; tail -40 gpurun_out/collect_final_tree.log | cut -c1-300
