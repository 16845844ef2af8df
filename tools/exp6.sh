#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SC_KSPLIT=64 timeout -k 10 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for k in 16 32 64; do
SC_KSPLIT=$k timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
done
