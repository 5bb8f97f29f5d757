#!/bin/bash
# round 6, session z: the whole GPU suite on the FINAL library with the oracle memo OFF (every oracle answer recomputed on the box's CPU cores)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6z}; mkdir -p $OUT
( time HIPETS_ORACLE_CACHE=0 timeout 2600 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 --ignore=tests/test_oracle_memo_pinned.py ) > $OUT/tests_nocache.log 2>&1
tail -12 $OUT/tests_nocache.log | cut -c1-200
echo done
