mkdir -p gpurun_out
log=gpurun_out/r06_ab_unobserved_rows_dynamic.log
: > $log
python scripts/ab_libs.py ab/lib_side.so ab/lib_udyn.so >> $log 2>&1
python scripts/ab_libs.py ab/lib_side.so ab/lib_udyn.so --state >> $log 2>&1
echo "== parity of the library with the unobserved entries' rows one store per k-th missing series of every model" >> $log
METRAN_HIP_LIBRARY=$PWD/ab/lib_udyn.so timeout 900 python -m pytest tests/test_dk_tape.py tests/test_gpu_property.py -q -m gpu -k "not 48 and not 19 and not 20 and not 17 and (dk_tape or 32x4 or 14x3)" 2>&1 | tail -5 >> $log
cat $log
