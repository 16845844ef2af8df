// nn_types.hpp -- kernel argument blocks of the network kernels (shared by host code and kernels).
#pragma once
#include <stdint.h>

#include "nn_kernels_layout.hpp"

namespace scnn {

typedef uint16_t bf16_t;  // storage

constexpr int HEAD = 256;      // head width (py/module.py:71,86)
constexpr int POL_PAD = 128;   // 73 policy channels padded to 8 MFMA column tiles
constexpr int FC1_N = 128;
constexpr int FC1_K = 64 * HEAD;  // 16384 (+7 meta handled in k_value_finish)
constexpr int PF = 3;             // weight prefetch distance (k-steps); kernels may read PF steps past a tensor

struct NetDev : NetLayout {
    const bf16_t* wb;  // packed bf16 GEMM operands
    const float* wf;   // fp32 per-channel parameters
};

struct TowerArgs {
    NetDev net;
    int n_pos;
    const int8_t* boards;     // [n][64][112]
    const int32_t* meta;      // [n][meta_stride]
    int meta_stride;
    const uint16_t* legal_idx;  // [n][224] or null
    const int32_t* n_legal;     // [n] or null
    float* prior;               // [n][224] or null
    float* logp;                // [n][4672] or null
    bf16_t* hval;               // [n][16384] value-head features in the tower's accumulator order (input of k_value_fc1; weights.hpp packs the FC rows to match)
    uint32_t* fc1_arrive;       // fused step kernel with value_head.ffn.0 inside the launch (step_kernels.hip): arrival counters, one per
                                // 64-position block (32 words apart), monotonic: +1 per workgroup and launch; else null
    uint32_t fc1_target;        // ... the count that says "every workgroup of the block has published its feature row in THIS launch"
    float* vpart;               // ... the split-K partials [64][n][128] that launch writes
    int fc1_acquire;            // ... 1: agent-scope acquire behind the poll as well (more than one workgroup per CU)
    float* dbg;                 // optional: [n][64][C] residual stream dump
    int dbg_stage;              // -1: none; 0: after stem; b>=1: after block b; 1000: final latent
};

struct Fc1Args {
    NetDev net;
    int n_pos, ksplit;
    const bf16_t* hval;
    float* vpart;
};

struct VfinArgs {
    NetDev net;
    int n_pos, ksplit;
    const float* vpart;
    const int32_t* meta;
    int meta_stride;
    float* value;
};

}  // namespace scnn
