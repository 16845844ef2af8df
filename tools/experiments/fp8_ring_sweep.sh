#!/bin/bash
# sweep ring geometries of the fp8 C=128 tower (experiment builds); prints conv cycles + total
cd $GRAFT_REPO_ROOT
for cfg in "6 3 3" "6 3 2" "3 3 3" "2 1 2" "4 2 2" "6 3 6" "12 9 3" "9 9 3"; do
  set -- $cfg
  SC_EXP_TAG=_s SC_EXP_DEFS="-DSC_T8_RS=$1 -DSC_T8_TPI=$2 -DSC_T8_AB=$3" bash tools/build_exp.sh > /dev/null 2>&1 || { echo "RS=$1 TPI=$2 AB=$3 build failed"; continue; }
  echo "== RS=$1 TPI=$2 AB=$3"
  SC_PREC=fp8 SC_DBG_C=128 SC_DBG_N=256 SC_ENGINE_LIB=smart-chess-rust_amd/lib_exp_s/libsc_engine.so python tools/dbg_tower.py 2>&1 | grep -E "total|conv1 |conv2 |stem: conv"
  SC_PREC=fp8 SC_ENGINE_LIB=smart-chess-rust_amd/lib_exp_s/libsc_engine.so python tools/tower_time.py 128 10 256 2>&1 | tail -1
done
