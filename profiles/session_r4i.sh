#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4i; mkdir -p $OUT
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_trace.so python profiles/turn_trace.py > $OUT/turn_trace.json 2> $OUT/err.log
cat $OUT/turn_trace.json; tail -3 $OUT/err.log
