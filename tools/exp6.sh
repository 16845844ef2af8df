#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SC_TOWER_W8=1 timeout -k 10 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
SC_TOWER_W8=1 timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
SC_TOWER_W8=1 timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
