#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3i; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest -m gpu -q --maxfail=10 -p no:cacheprovider tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_batched_plans.py
for v in main prof; do
  if [ $v = main ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_$v.so; fi
  run variants_$v python profiles/kernel_variants.py
  grep -h '^{' $OUT/variants_$v.log | tail -1 > $OUT/variants_$v.json
done
echo done
