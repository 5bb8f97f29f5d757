#!/bin/bash
# round 3, GPU session E: the k-loop probe (profiles/microbench/kloop_probe), phase profile of the shape-specialised fused kernel,
# the fully-unrolled-k variant.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3e
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 4 $OUT/$name.log | tr '\n' ' ' | cut -c1-400)"; }
run kloop profiles/microbench/kloop_probe
for v in main prof unroll; do
  if [ $v = main ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_$v.so; fi
  run variants_$v python profiles/kernel_variants.py
  grep -h '^{' $OUT/variants_$v.log | tail -1 > $OUT/variants_$v.json
done
export HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_unroll.so
run tests_unroll python -m pytest -m gpu -q --maxfail=10 -p no:cacheprovider tests/test_gpu_rollout.py tests/test_gpu_device_mode.py -k "shape_specialised or replayed or persistent"
unset HIPETS_LIB
echo done
