mkdir -p gpurun_out
log=gpurun_out/r06_ab_dk_late_output_stores.log
: > $log
python scripts/ab_libs.py ab/lib_side.so ab/lib_dk_late.so >> $log 2>&1
python scripts/ab_libs.py ab/lib_side.so ab/lib_dk_late.so --state >> $log 2>&1
echo "== parity of the library with the backward pass's outputs stored one step late" >> $log
METRAN_HIP_LIBRARY=$PWD/ab/lib_dk_late.so timeout 900 python -m pytest tests/test_dk_tape.py tests/test_gpu_property.py -q -m gpu -k "not 48 and (dk_tape or 32x4 or 14x3)" 2>&1 | tail -5 >> $log
cat $log
