#!/bin/bash
# A short GPU-box session: a few targeted tests, kernel-variant timings, a short bench line.  bash profiles/session_quick.sh <tag> [pytest -k expr]
set -u
TAG=${1:-q}
KEXPR=${2:-shape_specialised or planet or device_mode_replayed}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "$KEXPR"
run variants python profiles/kernel_variants.py
run bench python bench.py --steps 20 --warmup 3 --no-cpu-baseline
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
grep -h '^{' $OUT/variants.log | tail -1 > $OUT/variants.json
echo done
