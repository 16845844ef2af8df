#!/bin/bash
# developer tool: occupancy-1 and occupancy-2 experiment builds, clock + time of the tower at 256 / 512 positions
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for occ in 1 2; do
  L=$R/smart-chess-rust_amd/lib_exp_occ$occ/libsc_engine.so
  for n in 256 512; do
    echo "== build SC_T32_OCC=$occ, $n positions"
    SC_ENGINE_LIB=$L SC_DBG_N=$n timeout -k 10 120 python tools/dbg_clock.py || exit 1
    SC_ENGINE_LIB=$L timeout -k 10 60 python tools/tower_time.py 128 10 $n || exit 1
  done
done
