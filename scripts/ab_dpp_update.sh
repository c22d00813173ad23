mkdir -p gpurun_out
log=gpurun_out/r06_ab_no_replica_stores.log
: > $log
python scripts/ab_libs.py ab/lib_side.so ab/lib_nodup.so >> $log 2>&1
python scripts/ab_libs.py ab/lib_side.so ab/lib_nodup.so --state >> $log 2>&1
bash scripts/ab_c4f.sh ab/lib_side.so ab/lib_nodup.so >> $log 2>&1
echo "== tape parity of the library without replica stores" >> $log
METRAN_HIP_LIBRARY=$PWD/ab/lib_nodup.so timeout 900 python -m pytest tests/test_dk_tape.py -q -m gpu -k "not 48 and not 19 and not 20 and not 17" 2>&1 | tail -3 >> $log
cat $log
