mkdir -p gpurun_out
log=gpurun_out/r06_ab_dpp_predict.log
: > $log
python scripts/ab_libs.py ab/lib_base.so ab/lib_dppall.so ab/lib_pred.so >> $log 2>&1
python scripts/ab_libs.py ab/lib_base.so ab/lib_dppall.so ab/lib_pred.so --state >> $log 2>&1
echo "== parity of the library with the broadcast prediction (tape tests, wide parity tests, the AOT wide groups of the sweep)" >> $log
METRAN_HIP_LIBRARY=$PWD/ab/lib_pred.so timeout 900 python -m pytest tests/test_dk_tape.py tests/test_gpu_property.py tests/test_hip_parity.py -q -m gpu -k "not 48 and not runtime and not 20 and (dk_tape or 32x4 or 14x3 or hip_parity)" 2>&1 | tail -15 >> $log
echo "== c4_full_sym through the default library of the tree (split filter with the broadcast update)" >> $log
python bench.py --no-cpu-baseline --no-live-traffic --only c4_full_sym 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(r['secondary']['c4_full_sym'])[:900])" >> $log 2>&1
cat $log
