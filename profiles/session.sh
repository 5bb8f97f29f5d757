#!/bin/bash
# One GPU-box session: smoke, GPU tests (new ones first), kernel-variant timings, microbenchmarks, a short bench line.
# Usage (from the repo root on the box): bash profiles/session.sh <tag> [steps...]; everything lands in gpurun_out/<tag>/
set -u
TAG=${1:-s1}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run tests_new python -m pytest tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -q --maxfail=30 -p no:cacheprovider
run tests_rest python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --deselect tests/test_gpu_device_mode.py --deselect tests/test_gpu_plans_full_size.py
run variants_default python profiles/kernel_variants.py
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_mw3.so run variants_mw3 python profiles/kernel_variants.py
run microbench ./profiles/microbench/mfma_valu_2wave
run bench python bench.py --steps 10 --warmup 2 --cpu-budget 15
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
