mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_tier.log 2>&1; tail -5 gpurun_out/r06_gpu_tier.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
python bench.py --gpus 1 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 1500 gpurun_out/r06_bench_default.json
