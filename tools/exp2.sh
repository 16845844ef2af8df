#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { env "$@" timeout -k 10 120 python bench.py --no-alt --steps 5 --warmup 1 --cpu-budget 0 $EXTRA 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', 'sims/s', d['value'], 'tower_ms', d['roofline']['avg_launch_ms'], 'ms/ply', d['ms_per_step'])"; }
for CH in 128 256; do
 for K in 1 2 4; do EXTRA="--channels $CH --groups $K" run X=1; done
done
for D in 300 800 2000 6000; do EXTRA="--channels 128" run SC_TOWER_DELAY=$D; done
for D in 2000 8000; do EXTRA="--channels 256" run SC_TOWER_DELAY=$D; done
EXTRA="--channels 128 --groups 2" run SC_TOWER_DELAY=800
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-alt --cpu-budget 0 --channels 128 > /dev/null 2>&1; cat $R/gpurun_out/prof_v4/*/*kernel_stats.csv | head -8; rm -f $R/gpurun_out/prof_v4/*/*kernel_trace.csv
