#!/bin/bash
# round 6, session aa: phase profile of the HEADLINE instances (cfg2, R = 3, both modes) from a -DHIPETS_LEAN_PROF=1 build of rollout_r3*.hip
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6aa}; mkdir -p $OUT
HIPETS_LIB=$PWD/profiles/variants/leanprof3.so PHASE_CASES=cfg2 timeout 300 python profiles/one_tile_phase_profile.py > $OUT/phase_cfg2.log 2>&1
grep -h '^{"lib"' $OUT/phase_cfg2.log | tail -1 > $OUT/cfg2_phase_profile.json; tail -c 1500 $OUT/phase_cfg2.log
echo done
