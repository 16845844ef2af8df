#!/bin/bash
# experiment: conv loops of the 10x128 towers with the weight stream and / or the image reads removed (results invalid,
# timing valid): which of the two streams keeps the loops below the matrix rate?
cd $GRAFT_REPO_ROOT
for defs in "" "-DSC_EXP_NOWLOAD" "-DSC_EXP_NOLDS" "-DSC_EXP_NOWLOAD -DSC_EXP_NOLDS"; do
  SC_EXP_TAG=_c SC_EXP_DEFS="$defs" bash tools/build_exp.sh > /dev/null 2>&1 || { echo "build failed: $defs"; continue; }
  for prec in bf16 fp8; do
    echo "== $prec defs='$defs'"
    SC_PREC=$prec SC_DBG_C=128 SC_DBG_N=256 SC_ENGINE_LIB=smart-chess-rust_amd/lib_exp_c/libsc_engine.so python tools/dbg_tower.py 2>&1 | grep -E "total|conv1 |conv2 "
  done
done
