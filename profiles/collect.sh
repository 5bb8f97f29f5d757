#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench command  -> per-kernel average durations
#   2. separate --pmc passes (never combined with trace domains other than kernel-trace): SQ busy/wait, LDS, and
#      FETCH_SIZE / WRITE_SIZE for the HBM traffic of the rollout kernel
# Outputs land in gpurun_out/prof_$TAG; profiles/summarize.py turns them into the committed summaries.
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-batched"
SHORT="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batched"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -o pmc -- $SHORT > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o pmc -- $SHORT > $OUT/pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $SHORT > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $SHORT > $OUT/pmc_write.log 2>&1
grep -h '"metric"' $OUT/bench_trace.log | tail -1 > $OUT/bench_line.json
find $OUT -name "*.csv" | head -30
