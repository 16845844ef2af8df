// step_kernels.hip -- the fused simulation-step kernel: one launch instead of two.
//
// k_mcts (one wavefront per game) and the network tower (one 4-wave workgroup per position) work on the SAME game on the
// SAME compute unit back to back: here the game's search wave is wave 0 of its tower workgroup.  It finishes the previous
// simulation (value-head tail, expand, backup -- dev_expand), selects the next leaf and encodes its planes into LDS
// (dev_select) -- inside the tower's prologue: every wave has already requested its first weights and the stem parameters,
// and the other three waves zero the LDS image meanwhile; then all four run the tower on the planes in LDS.  Saved per simulation step: one launch ramp, the tower's cold prologue behind a kernel boundary, and the 7 KB
// round trip of the planes through HBM.  A game whose leaf needs no network (terminal position, idle slot) skips the tower.
// Bit-identical to k_mcts + k_tower32 (same device functions): tests/test_gpu_parity*.py run through this path; the
// unfused pair remains for the synthetic evaluators, the final flush and the timed samples of the bench.
//
// The search arithmetic must not be contracted into FMAs (src/mcts.rs:69-75 is plain f32) while the tower's epilogues are
// written for contraction: this unit is compiled like nn_kernels.hip (contraction on), and the search functions carry
// `#pragma clang fp contract(off)` in their bodies and use raw hardware transcendentals, so that they compile to the same
// instructions here and in mcts_kernels.hip (checked: k_mcts is identical with and without -ffp-contract=off).
#define SC_NO_KERNELS   // device functions only: the kernels of these headers live in their own translation units
#include "mcts_kernels.hpp"
#include "nn_kernels.hpp"
#include "nn_tower32.hpp"

#include "launchers.hpp"
#include "tower_config.hpp"

namespace scstep {

template <class P, int C, int RS, int TPI, int AB>
__global__ __launch_bounds__(256, 1) void k_step(scnn::TowerArgs A, sc::SpParams p, int do_expand) {
    __shared__ __attribute__((aligned(16))) int8_t s_stage[7168];
    __shared__ sc::move_t s_moves[sc::MAXC];
    __shared__ sc::Position s_pos;
    __shared__ sc::Position s_hist[8];
    __shared__ uint16_t s_ps[sc::DEPTH_LDS];
    __shared__ sc::HelperBox s_box;
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63;
    // wave 0: the game's search, wave 1: its helper (encodes the leaf's planes while wave 0 generates the moves); both run
    // inside the tower's prologue (tower_body, Pre), after every wave has requested its first weights
    if (threadIdx.x == 0) s_box.state = 0;   // LDS starts with whatever the previous workgroup left: clear the mailbox ...
    __syncthreads();                         // ... before wave 1 can poll it or wave 0 post to it
    auto search = [&](int wave) -> bool {
        if (wave == 1) {
            sc::dev_encode_helper(p, g, lane, &s_box, s_stage, &s_pos, s_ps, s_hist);
            return true;
        }
        sc::GameCtl cs_pre{};
        bool cs_pre_valid = false;
        if (do_expand) {
            sc::dev_expand(p, g, lane, &s_pos, cs_pre, cs_pre_valid);
            __builtin_amdgcn_wave_barrier();
        }
        return sc::dev_select<false>(p, g, lane, s_stage, s_moves, &s_pos, s_ps, s_hist, cs_pre, cs_pre_valid, &s_box);
    };
    scnn::tower_body<P, C, RS, TPI, AB>(A, g, s_stage, search);
}

}  // namespace scstep

namespace scl {
#define K_STEP(P, C, RS, TPI, AB) scstep::k_step<scnn::P, C, RS, TPI, AB>
const char* step_init() {
    const void* kn[4] = {reinterpret_cast<const void*>(&K_STEP(PrecBF16, 128, SC_T32_RS, SC_T32_TPI, SC_T32_AB)),
                         reinterpret_cast<const void*>(&K_STEP(PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32_AB)),
                         reinterpret_cast<const void*>(&K_STEP(PrecFP8, 128, SC_T8_RS, SC_T8_TPI, SC_T8_AB)),
                         reinterpret_cast<const void*>(&K_STEP(PrecFP8, 256, SC_T8W_RS, SC_T8W_TPI, SC_T8W_AB))};
    for (int i = 0; i < 4; i++) {
        hipError_t e = hipFuncSetAttribute(kn[i], hipFuncAttributeMaxDynamicSharedMemorySize, scnn::tower32_lds_bytes(i & 1 ? 256 : 128, i >= 2 && i < 4));
        if (e != hipSuccess) return hipGetErrorString(e);
    }
    return nullptr;
}
// fused workgroups that can be resident on one CU (registers + LDS, asked of the runtime): the engine uses the fused form
// only when every game's workgroup is resident at once
int step_blocks_per_cu(const scnn::NetLayout& net) {
    int n = 0;
    hipError_t e;
    if (net.fp8 && net.C == 128)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K_STEP(PrecFP8, 128, SC_T8_RS, SC_T8_TPI, SC_T8_AB), 256, scnn::tower32_lds_bytes(128, true));
    else if (net.fp8)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K_STEP(PrecFP8, 256, SC_T8W_RS, SC_T8W_TPI, SC_T8W_AB), 256, scnn::tower32_lds_bytes(256, true));
    else if (net.C == 128)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K_STEP(PrecBF16, 128, SC_T32_RS, SC_T32_TPI, SC_T32_AB), 256, scnn::tower32_lds_bytes(128));
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K_STEP(PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32_AB), 256, scnn::tower32_lds_bytes(256));
    return e == hipSuccess && n > 0 ? n : 1;
}
void step(const scnn::TowerArgs& a, const sc::SpParams& p, int do_expand, hipStream_t s) {
    const dim3 grid(p.n_slots), block(256);
    if (a.net.fp8 && a.net.C == 128)
        hipLaunchKernelGGL((K_STEP(PrecFP8, 128, SC_T8_RS, SC_T8_TPI, SC_T8_AB)), grid, block, scnn::tower32_lds_bytes(128, true), s, a, p, do_expand);
    else if (a.net.fp8)
        hipLaunchKernelGGL((K_STEP(PrecFP8, 256, SC_T8W_RS, SC_T8W_TPI, SC_T8W_AB)), grid, block, scnn::tower32_lds_bytes(256, true), s, a, p, do_expand);
    else if (a.net.C == 128)
        hipLaunchKernelGGL((K_STEP(PrecBF16, 128, SC_T32_RS, SC_T32_TPI, SC_T32_AB)), grid, block, scnn::tower32_lds_bytes(128), s, a, p, do_expand);
    else
        hipLaunchKernelGGL((K_STEP(PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32_AB)), grid, block, scnn::tower32_lds_bytes(256), s, a, p, do_expand);
}
}  // namespace scl
