// nn_kernels_layout.hpp -- element offsets of the packed network parameters (host + device).
#pragma once
#include <stddef.h>
#include <stdint.h>
namespace scnn {
struct NetLayout {
    int n_blocks, C;
    int tower32;  // 1: trunk/head convs packed for k_tower32 (32x32x16 A fragments), 0: the experiment-only pixel-major kernel (tools/experiments/nn_tower16.hpp, 16x16x32 B fragments)
    int fp8;      // 1: the conv weights (stem, blocks, head convs) are OCP e4m3 fragments for v_mfma_scale_f32_32x32x64_f8f6f4 with
                  // per-output-channel power-of-two scales (f_scales); the SE / value FC layers stay bf16
    // element offsets (units of 2 bytes) into the GEMM-operand blob (MFMA fragment order)
    size_t o_stem, o_blocks, blk_stride_b, o_vconv, o_pconv1, o_pconv2, o_fc1;
    // element offsets into the fp32 blob (per-channel parameters, logical channel order)
    size_t f_stem, f_blocks, blk_stride_f, f_vhead, f_phead1, f_phead2, f_fc1b, f_fc1m, f_fc2w, f_fc2b;
    size_t f_scales;   // fp8: E8M0 scale bytes as ints [conv][8 tiles][64 lanes] (conv 0 stem, 1 + 2b / 2 + 2b block b, then the heads)
};
}  // namespace scnn
