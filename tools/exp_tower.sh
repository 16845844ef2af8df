#!/bin/bash
# experiment: tower knobs (developer tool)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for CH in 256 128; do
 for RING in 4 8; do
  for ST in 0 1 4; do
   if [ $CH = 128 ] && [ $RING = 8 ]; then continue; fi
   SC_TOWER_RING=$RING SC_TOWER_STAGGER=$ST timeout -k 10 120 python bench.py --channels $CH --no-alt --steps 5 --warmup 1 --cpu-budget 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C=$CH ring=$RING stagger=$ST', 'sims/s', d['value'], 'tower_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'ms/ply', d['ms_per_step'])"
  done
 done
done
