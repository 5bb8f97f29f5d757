#!/bin/bash
# round 6, session y: the narrow PlaNet form over population sizes -- is its weight stream bound per CU or chip-wide?
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6y}; mkdir -p $OUT
timeout 300 python profiles/planet_narrow_probe.py > $OUT/probe.log 2>&1; tail -1 $OUT/probe.log
echo done
