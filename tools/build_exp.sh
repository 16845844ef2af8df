#!/bin/bash
# developer tool: experiment build of libsc_engine.so (-DSC_EXP knobs) into smart-chess-rust_amd/lib_exp/
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/smart-chess-rust_amd/csrc; O=$R/smart-chess-rust_amd/lib_exp$SC_EXP_TAG; mkdir -p $O
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSC_EXP $SC_EXP_DEFS"
$H -ffp-contract=off -c $C/mcts_kernels.hip -o $O/mcts.o 
$H -mllvm -amdgpu-mfma-vgpr-form=1 -c $C/nn_kernels.hip -o $O/nn.o 
$H -mllvm -amdgpu-mfma-vgpr-form=1 -c $C/step_kernels.hip -o $O/step.o
$H -c $C/engine.hip -o $O/engine.o 
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libsc_engine.so $O/mcts.o $O/nn.o $O/step.o $O/engine.o -Wl,-rpath,/opt/rocm/lib
rm -f $O/*.o; ls -la $O
