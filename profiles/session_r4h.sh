#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4h; mkdir -p $OUT
python profiles/cfg4p_probe.py > $OUT/cfg4p_main.json 2> $OUT/err1.log
HIPETS_LIB=$PWD/mbrl-lib_amd/hipets/libhipets_half.so python profiles/cfg4p_probe.py > $OUT/cfg4p_half.json 2> $OUT/err2.log
cat $OUT/cfg4p_main.json $OUT/cfg4p_half.json; tail -2 $OUT/err2.log
