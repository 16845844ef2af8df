// nn_kernels_layout.hpp -- element offsets of the packed network parameters (host + device).
#pragma once
#include <stddef.h>
#include <stdint.h>
namespace scnn {
struct NetLayout {
    int n_blocks, C;
    int tower32;  // 1: trunk/head convs packed for k_tower32 (32x32x16 A fragments), 0: the experiment-only pixel-major kernel (tools/experiments/nn_tower16.hpp, 16x16x32 B fragments)
    // element offsets into the bf16 blob (MFMA B-fragment packed GEMM operands)
    size_t o_stem, o_blocks, blk_stride_b, o_vconv, o_pconv1, o_pconv2, o_fc1;
    // element offsets into the fp32 blob (per-channel parameters, logical channel order)
    size_t f_stem, f_blocks, blk_stride_f, f_vhead, f_phead1, f_phead2, f_fc1b, f_fc1m, f_fc2w, f_fc2b;
};
}  // namespace scnn
