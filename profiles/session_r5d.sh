#!/bin/bash
# round 5: cross-barrier prefetch of the one-tile k-split instances (HIPETS_KS_PREFETCH) against the build without it and against no
# k-split, same box; phase profile of the fused one-tile instance; the tests that run one-tile instances
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5d; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run small_default python profiles/small_batch_probe.py
HIPETS_LIB=$PWD/profiles/variants/nopf.so run small_nopf python profiles/small_batch_probe.py
HIPETS_LIB=$PWD/profiles/variants/noks.so run small_noks python profiles/small_batch_probe.py
HIPETS_LIB=$PWD/profiles/variants/prof.so run small_prof python profiles/small_batch_probe.py
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests/test_gpu_rollout.py tests/test_gpu_closed_forms.py tests/test_gpu_device_mode.py tests/test_gpu_planning.py -m gpu -q --maxfail=40 -p no:cacheprovider
echo done
