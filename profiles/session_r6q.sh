#!/bin/bash
# round 6, session q: paired Philox draws in the fused tail (one block per lane and PAIR of units): the whole GPU suite, bench line, cfg4' sizes
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6q}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1200} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 -x
run bench python bench.py --no-cpu-baseline
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
PROBE_RS=0 run cfg4p_iterations python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations.log | tail -1 > $OUT/cfg4p_iterations.json
run headline python profiles/headline_probe.py
echo done
HIPETS_LIB=$PWD/profiles/variants/noshare.so run headline_noshare python profiles/headline_probe.py
HIPETS_LIB=$PWD/profiles/variants/noshare.so PROBE_RS=0 run cfg4p_iterations_noshare python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations_noshare.log | tail -1 > $OUT/cfg4p_iterations_noshare.json
echo done2
