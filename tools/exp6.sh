#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
SC_ENGINE_LIB=$R/smart-chess-rust_amd/lib_exp/libsc_engine.so timeout -k 10 100 python tools/dbg_tower.py
timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10
