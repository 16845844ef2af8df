#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box via gpurun; summaries are copied to profiles/ afterwards
# by tools/profile_summarize.py).  Counters are collected in their own passes (no trace domains besides kernel-trace).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
CH=${2:-128}
EXTRA=${3:-}     # further bench.py arguments of the profiled configuration, e.g. "--precision fp8" or "--blocks 20 --rollout 800"
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --repeats 1 --no-alt --cpu-budget 0 --channels $CH $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c$CH -- python3 $R/bench.py $ARGS > $OUT/bench_stats_c$CH.json 2> $OUT/stats_c$CH.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c$CH -- python3 $R/bench.py --steps 1 --warmup 1 --repeats 1 --no-alt --cpu-budget 0 --channels $CH $EXTRA > /dev/null 2> $OUT/pmc_fetch_c$CH.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c$CH -- python3 $R/bench.py --steps 1 --warmup 1 --repeats 1 --no-alt --cpu-budget 0 --channels $CH $EXTRA > /dev/null 2> $OUT/pmc_write_c$CH.err
echo "write pass done"
find $OUT -name "*.csv" | head -20
# keep the merge small: drop per-dispatch traces, keep stats + counter csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
du -sh $OUT
# MFMA-pipe utilisation of the kernels (SQ + GRBM counters, their own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_c$CH -- python3 $R/bench.py --steps 1 --warmup 1 --repeats 1 --no-alt --cpu-budget 0 --channels $CH $EXTRA > /dev/null 2> $OUT/pmc_mfma_c$CH.err
echo "mfma pass done"
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
