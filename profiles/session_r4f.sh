#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4f; mkdir -p $OUT
python profiles/batched_small_probe.py > $OUT/batched_small.json 2> $OUT/batched_small.err
tail -3 $OUT/batched_small.err
echo done
