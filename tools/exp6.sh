#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2 3; do
timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
SC_ENGINE_LIB=$R/smart-chess-rust_amd/lib_exp_reslds/libsc_engine.so timeout -k 10 120 python bench.py --cpu-budget 0 --steps 10 --no-alt
done
