#!/bin/bash
# developer tool: power / clock digest of the burn experiment, the tower alone and the bench configurations (one gpurun call)
R=$GRAFT_REPO_ROOT; cd $R
(rocm-smi --showmaxpower; rocm-smi --showpowercap 2>&1 | head -5; rocm-smi -M 2>&1 | head -8; amd-smi static --limit 2>&1 | head -30) > gpurun_out/pw_caps.txt 2>&1
tools/power_sample.sh pw_burn1 tools/experiments/mfma_burn.bin 1.5 1 0
tools/power_sample.sh pw_burn2 tools/experiments/mfma_burn.bin 1.5 2 0
tools/power_sample.sh pw_burn1z tools/experiments/mfma_burn.bin 1.5 1 1
SC_TT_STEPS=12000 tools/power_sample.sh pw_tower256 python3 tools/tower_time.py 128 10 256
SC_TT_STEPS=8000 tools/power_sample.sh pw_tower512 python3 tools/tower_time.py 128 10 512
tools/power_sample.sh pw_bench_bf16 python3 bench.py --steps 40 --warmup 2 --no-alt --cpu-budget 0
tools/power_sample.sh pw_bench_fp8 python3 bench.py --steps 40 --warmup 2 --no-alt --cpu-budget 0 --precision fp8
tools/power_sample.sh pw_bench_fp8x512 python3 bench.py --steps 40 --warmup 2 --no-alt --cpu-budget 0 --precision fp8 --games 512
tools/power_sample.sh pw_bench_c256 python3 bench.py --steps 20 --warmup 2 --no-alt --cpu-budget 0 --channels 256
