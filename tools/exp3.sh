#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { env "$@" timeout -k 10 120 python bench.py --no-alt --steps 5 --warmup 1 --cpu-budget 0 $EXTRA 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', 'sims/s', d['value'], 'tower_ms', d['roofline']['avg_launch_ms'], 'ms/ply', d['ms_per_step'])"; }
EXTRA="--channels 128" run SC_TOWER_RING=4
EXTRA="--channels 128" run SC_TOWER_RING=12
EXTRA="--channels 128" run SC_TOWER_RING=12 SC_TOWER_STAGGER=1
EXTRA="--channels 128" run SC_TOWER_RING=4 SC_TOWER_STAGGER=1
EXTRA="--channels 256" run SC_TOWER_RING=4
cd /tmp && export TMPDIR=/tmp && SC_TOWER_RING=12 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v8 -- python3 $R/bench.py --steps 3 --warmup 1 --no-alt --cpu-budget 0 --channels 128 > /dev/null 2>&1; head -4 $R/gpurun_out/prof_v8/*/*kernel_stats.csv; rm -f $R/gpurun_out/prof_v8/*/*kernel_trace.csv
