#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
