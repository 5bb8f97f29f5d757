#!/bin/bash
# round 3, GPU session B: the fused output-layer tail (KSpec::FUSE) -- parity of the lean instances, then A/B timings against
# the same tree built with -DHIPETS_FUSE_TAIL=0 (libhipets_nofuse.so).  bash profiles/session_r3b.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3b
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 4 $OUT/$name.log | tr '\n' ' ' | cut -c1-400)"; }
run tests_fuse python -m pytest -m gpu -q --maxfail=8 -p no:cacheprovider tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_stock_configs.py "tests/test_gpu_plans_full_size.py::test_fused_cem_plan_cfg2_replayed_through_oracle" tests/test_gpu_batched_plans.py
run variants_fuse python profiles/kernel_variants.py
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_nofuse.so run variants_nofuse python profiles/kernel_variants.py
run bench python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
grep -h '^{' $OUT/variants_fuse.log | tail -1 > $OUT/variants_fuse.json
grep -h '^{' $OUT/variants_nofuse.log | tail -1 > $OUT/variants_nofuse.json
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
