#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the bench command, headline (DEVICE) mode and FAST mode  -> per-kernel average durations
#   2. separate --pmc passes (never combined with trace domains other than kernel-trace): SQ busy / wait / MFMA, LDS, instruction
#      cache, and FETCH_SIZE / WRITE_SIZE for the HBM traffic of the rollout kernel, for both modes
#   3. --kernel-trace --stats of every other BASELINE configuration (profiles/other_configs.py --only ...)
# Outputs land in gpurun_out/prof_$TAG; profiles/summarize.py turns them into the committed summaries.
set -u
TAG=${1:-r2}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
for MODE in device fast; do
  BENCH="python bench.py --mode $MODE --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
  SHORT="python bench.py --mode $MODE --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -o bench -- $BENCH > $OUT/bench_trace_$MODE.log 2>&1
  grep -h '"metric"' $OUT/bench_trace_$MODE.log | tail -1 > $OUT/bench_line_$MODE.json
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq_$MODE -o pmc -- $SHORT > $OUT/pmc_sq_$MODE.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds_$MODE -o pmc -- $SHORT > $OUT/pmc_lds_$MODE.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODE -o pmc -- $SHORT > $OUT/pmc_fetch_$MODE.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODE -o pmc -- $SHORT > $OUT/pmc_write_$MODE.log 2>&1
done
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_icache_fast -o pmc -- python bench.py --mode fast --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pmc_icache_fast.log 2>&1
for c in cfg1_cartpole cfg4_humanoid_truncated_obs cfg4_humanoid_v4_obs376 cfg5_cheetah_run planet cfg1_cem_plan cfg4_icem_plan cfg5_mppi_plan; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_$c -o t -- python profiles/other_configs.py --only $c --mode device --reps 5 > $OUT/cfg_$c.log 2>&1
done
# the workloads the reference ships (round 4): kernel trace of their rollouts, both randomness modes
for W in stock_halfcheetah stock_cartpole stock_pusher; do
  for MODE in device fast; do
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_${W}_$MODE -o t -- python profiles/stock_workloads.py --only $W --mode $MODE --reps 10 --no-plans > $OUT/cfg_${W}_$MODE.log 2>&1
  done
done
python profiles/stock_workloads.py --sweep-r --generic > $OUT/stock_workloads.json 2> $OUT/stock_workloads.err
# the separately reported bf16x3 arithmetic mode: kernel trace of cfg2 rollouts in both randomness modes, L2 hit / miss counters
for MODE in device fast; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_cfg2_bf16x3_$MODE -o t -- python profiles/precision_probe.py --precision bf16x3 --mode $MODE --reps 10 > $OUT/cfg_cfg2_bf16x3_$MODE.log 2>&1
done
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc_bf16x3_device -o pmc -- python profiles/precision_probe.py --precision bf16x3 --mode device --reps 3 > $OUT/pmc_tcc_bf16x3_device.log 2>&1
python profiles/other_configs.py --reps 6 > $OUT/other_configs.json 2> $OUT/other_configs.err
python profiles/small_batch_probe.py > $OUT/small_batches.json 2> $OUT/small_batches.err
find $OUT -name "*_kernel_stats.csv" | head -40
find $OUT -name "*.csv" -size +2M -delete   # the per-dispatch traces are large; the stats summaries are what gets committed
