#!/bin/bash
# round 3, GPU session C: fused tail (three-stage, branch-free) + buffer-load weight fragments + compile-time LDS stride + SPL output
# layer: parity of the main build, then kernel timings of the build variants (HIPETS_LIB).  bash profiles/session_r3c.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3c
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1200} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 4 $OUT/$name.log | tr '\n' ' ' | cut -c1-400)"; }
run tests_main python -m pytest -m gpu -q --maxfail=10 -p no:cacheprovider tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_closed_forms.py tests/test_gpu_planet.py tests/test_gpu_bf16x3.py tests/test_gpu_batched_plans.py tests/test_gpu_planning.py tests/test_gpu_c_abi.py tests/test_gpu_closed_loop.py "tests/test_gpu_plans_full_size.py::test_fused_cem_plan_cfg2_replayed_through_oracle"
for v in main nofuse nonop nobuf prof; do
  if [ $v = main ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_$v.so; fi
  run variants_$v python profiles/kernel_variants.py
  grep -h '^{' $OUT/variants_$v.log | tail -1 > $OUT/variants_$v.json
done
unset HIPETS_LIB
run bench python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
