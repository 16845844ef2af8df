#!/bin/bash
# developer tool: a production-flags build of the library with extra compiler arguments, for same-box A/B runs (tools/ab_r02.py)
# usage: tools/build_variant.sh <tag> "<extra hipcc args>"   -> smart-chess-rust_amd/lib_<tag>/libsc_engine.so
set -e
R=/root/repo; C=$R/smart-chess-rust_amd/csrc; O=$R/smart-chess-rust_amd/lib_$1; mkdir -p $O
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2"
$H -ffp-contract=off -c $C/mcts_kernels.hip -o $O/mcts.o &
$H -mllvm -amdgpu-mfma-vgpr-form=1 -c $C/nn_kernels.hip -o $O/nn.o &
$H -mllvm -amdgpu-mfma-vgpr-form=1 -c $C/step_kernels.hip -o $O/step.o &
$H -c $C/engine.hip -o $O/engine.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libsc_engine.so $O/mcts.o $O/nn.o $O/step.o $O/engine.o -Wl,-rpath,/opt/rocm/lib
rm -f $O/*.o
