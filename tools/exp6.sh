#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for s in 1 8 64 1 8 64; do
timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt --timing-stride $s
done
