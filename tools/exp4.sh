#!/bin/bash
# experiments: ring depth and weight-window (L2-hot) bounds of the tower
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== base"; timeout -k 10 100 python tools/dbg_tower.py
echo "== ring 12 (C=128)"; SC_TOWER_RING=12 timeout -k 10 100 python tools/dbg_tower.py
export SC_ENGINE_LIB=$R/smart-chess-rust_amd/lib_exp/libsc_engine.so
echo "== exp lib, no mask"; timeout -k 10 100 python tools/dbg_tower.py
echo "== exp lib, 32 KB window"; SC_EXP_WAND=0x7fff timeout -k 10 100 python tools/dbg_tower.py
echo "== exp lib, 32 KB window ring 12"; SC_TOWER_RING=12 SC_EXP_WAND=0x7fff timeout -k 10 100 python tools/dbg_tower.py
