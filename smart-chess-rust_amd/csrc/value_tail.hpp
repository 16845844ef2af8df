// value_tail.hpp -- the tail of the value head for ONE position, run by ONE wavefront (py/module.py:95-106,147-149):
// split-K partials of value_head.ffn.0 + bias + the 7 meta columns, ReLU, Linear 128->1, tanh, times (2*turn-1).
//
// There are two callers and they must agree to the last bit, because `Game::predict` (src/backends/torch.rs:89-146) and
// the value the search backs up are the same number in the reference (src/mcts.rs:149-152):
//   * k_value_finish (nn_kernels.hpp)   -- the L-predict entry points of the C ABI (sc_forward_batch / sc_predict_batch)
//   * dev_expand (mcts_kernels.hpp)     -- the search wave of k_mcts / k_step, on the partials of the previous launch
// Both fill a ValueTail and call value_tail_compute: one body, carrying its own `fp contract(off)`, so that it compiles to
// the same instructions in every translation unit whatever that unit's contraction setting (the library expansion of tanhf
// follows the setting in force where it is inlined: a k_value_finish with its own copy of this arithmetic differed from the
// search's value by one ulp on some positions -- tests/test_gpu_netloop.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scvt {

struct ValueTail {
    float2 acc[64];      // split-K partials of output columns 2*lane, 2*lane+1 (entries >= ksplit are not read)
    float2 bias, w2, wm[7];
    int32_t meta[7];
    float fc2b;
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// float sum over the wave in a FIXED order: four DPP steps inside each row of 16 lanes, then the four row sums
__device__ __forceinline__ float wave_sum_fixed(float v) {
#pragma clang fp contract(off)
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}

// ksplit is 32 or 64 (engine.hip).  Wave-uniform result.
__device__ __forceinline__ float value_tail_compute(const ValueTail& t, const int ksplit) {
#pragma clang fp contract(off)
    float m[7];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        // meta is fed to the net as bf16 (src/backends/torch.rs:120-123): round to nearest even
        uint32_t u = __builtin_bit_cast(uint32_t, (float)t.meta[k]);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        m[k] = __builtin_bit_cast(float, u);
    }
    float s0 = t.bias.x, s1 = t.bias.y;
#pragma unroll
    for (int ks = 0; ks < 32; ks++) {
        s0 += t.acc[ks].x;
        s1 += t.acc[ks].y;
    }
    if (ksplit > 32) {
#pragma unroll
        for (int ks = 32; ks < 64; ks++) {
            s0 += t.acc[ks].x;
            s1 += t.acc[ks].y;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; k++) {
        s0 += m[k] * t.wm[k].x;
        s1 += m[k] * t.wm[k].y;
    }
    s0 = s0 < 0.f ? 0.f : s0;   // ReLU that keeps a NaN (torch.relu does; `s > 0 ? s : 0` would swallow it)
    s1 = s1 < 0.f ? 0.f : s1;
    float part = s0 * t.w2.x + s1 * t.w2.y;
    part = wave_sum_fixed(part);
    float v = tanhf(part + t.fc2b);
    return v * (float)(t.meta[0] * 2 - 1);
}

}  // namespace scvt
