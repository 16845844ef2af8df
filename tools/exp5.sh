#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for t in _now _noa _none; do
export SC_ENGINE_LIB=$R/smart-chess-rust_amd/lib_exp$t/libsc_engine.so
echo "== $t"; timeout -k 10 100 python tools/dbg_tower.py
done
