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

// value_head.ffn.0 inside the step launch (A.fc1_arrive != null; the engine chooses it when the slots fill whole 64-position
// blocks and every workgroup of the launch is resident).  The layer is the network's only cross-position GEMM: tile
// (block of 64 positions, K chunk of 256) needs the feature rows of 64 workgroups.  Each workgroup publishes its row right
// behind the value conv (write-through stores, then one arrival per workgroup on its block's counter: tower_body), runs the
// policy head, and then computes tile (its block, K chunk = its index in the block): one lane polls the counter, the rows come
// in past the L1 (sc1 loads: hand-off form "row 1" of MI355X_MICROARCH.md; with two workgroups per CU, which that table does
// not cover, an agent-scope acquire as well).  The tile's weights, a counter reading and -- where an earlier reading already
// showed the block complete -- the rows themselves are requested by the tower under its softmax (Fc1Hand, nn_tower32.hpp).  The partials go to the same array k_value_fc1
// writes, for the next launch's value tail: bit-identical (same tile code, nn_kernels.hpp).  What it saves is a kernel
// whose 9 us are mostly its cold start.  No workgroup waits for a later block, and workgroups start in index order, so the
// wait ends even when not every workgroup is resident; it is bounded all the same (ERR_HANDOFF_TIMEOUT, ~0.2 s).
__device__ __forceinline__ void fc1_tail(const scnn::TowerArgs& A, const sc::SpParams& p, const int g, const bool ran, scnn::Fc1Hand& fh) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = g >> 6, ks = g & 63;
    uint32_t* ctr = A.fc1_arrive + mb * 32;
#ifdef SC_EXP   // experiment builds: stamps of the tail's phases (100 MHz), read by tools/dbg_tail.py
#define TSTAMP(k) do { if (p.dbg_cycles && tid == 0) p.dbg_cycles[(size_t)g * 32 + 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(k)
#endif
    TSTAMP(0);
    // a workgroup whose leaf needed no network has nothing to publish (its row keeps the last evaluation's features; the
    // partials computed from it are never read), but the others count on its arrival
    // (a workgroup that ran the network requested the tile's weights and a first reading of the counter under its softmax)
    if (!ran) {
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scnn::fc1_wload(fh.w, A.net, ks, 64, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    const scnn::Fc1W& w = fh.w;
    const bool have_a = ran && fh.have_a;   // (uniform over the workgroup)
#ifdef SC_EXP
    if (p.dbg_cycles && tid == 0) p.dbg_cycles[(size_t)g * 32 + 14] = have_a;
    if (have_a) { TSTAMP(1); TSTAMP(2); }
#endif
    if (tid == 0 && !have_a) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        uint32_t seen = ran ? fh.early : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int32_t)(seen - A.fc1_target) < 0) {
            __builtin_amdgcn_s_sleep(4);
            const uint64_t waited = __builtin_amdgcn_s_memrealtime() - t0;
            if (waited > 20000000ull) {
                atomicOr(&p.cnt->err, sc::ERR_HANDOFF_TIMEOUT);
                break;
            }
            // A hand-off of this handle has already timed out (this launch or an earlier one): its results are invalid anyway and
            // the host refuses the handle as soon as it looks (SC_ERR_HANDOFF).  Every later launch waiting its own 0.2 s would
            // turn a queue of thousands of launches into a stall of many minutes: after 1 ms of waiting, give up as well.
            if (waited > 100000ull && (__hip_atomic_load(&p.cnt->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & sc::ERR_HANDOFF_TIMEOUT)) break;
            seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        TSTAMP(1);
        if (A.fc1_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(2);
    }
    __syncthreads();   // the poll has matched (and the tower's last LDS reads are done: the tile is staged over its image)
    scnn::bf16_t* s_a = reinterpret_cast<scnn::bf16_t*>(scnn::g_smem);
    if (!have_a) scnn::fc1_load_a(fh.a, A.hval, mb, ks, tid);
    scnn::fc1_put_a(s_a, fh.a, tid);
    __syncthreads();
    TSTAMP(3);
    scnn::fc1_mma_store(w, s_a, A.vpart, A.n_pos, mb, ks, 64, wave, lane);
    TSTAMP(4);
#ifdef SC_EXP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP(5);
#endif
}

// STAMPS: the instantiation with the launch's phase stamps (bench.py's per-phase split; the headline's kernel only).  Compiled into
// every launch they cost 0.5 % (same-box A/B), so the engine launches this variant only while stamps are switched on.
template <class P, int C, int RS, int TPI, int AB, bool STAMPS = false>
__global__ __launch_bounds__(256, 1) void k_step(scnn::TowerArgs A, sc::SpParams p, int do_expand) {
    // (Round 3 tried the 7 KB plane staging inside the tower's dynamic LDS -- the policy head's image area, free until the heads -- to
    // get the bf16 workgroup from 83.5 to 76 KB, two per CU: 512 bf16 games then run the one-launch step, 2.47 M simulations/s against
    // 2.28 M in three launches.  But the 256-game headline lost 2.3 % (same-box A/B against this form; the compiler can no longer tell
    // the staging area from the tower's other LDS traffic), and two interleaved groups serve 512 games better anyway: reverted.)
    __shared__ __attribute__((aligned(16))) int8_t s_stage[7168];
    __shared__ sc::move_t s_moves[sc::MAXC];
    __shared__ sc::Position s_pos;
    __shared__ sc::Position s_hist[8];
    __shared__ uint16_t s_ps[sc::DEPTH_LDS];
    __shared__ sc::HelperBox s_box;
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63;
#ifdef SC_EXP
    const long long t_entry = clock64();   // experiment builds: the kernel's first instruction (tools/dbg_expand.py)
#endif
    // the two argument blocks: 684 bytes = 11 cache lines (kernarg_prefetch, nn_kernels.hpp)
    scnn::kernarg_prefetch<sizeof(scnn::TowerArgs) + sizeof(sc::SpParams) + sizeof(int)>();
#ifdef SC_EXP
    if (p.dbg_cycles && threadIdx.x == 0) p.dbg_cycles[(size_t)g * 32 + 21] = t_entry;
#endif
    // phase stamps of the launch (sc_selfplay_debug_cycles; bench.py's per-phase split): 100 MHz wall clock at kernel entry
    // [24], when the search wave has its leaf [25], at the end of the network [26] and of the value-FC tile [27]
#define PHASE_STAMP(k, cond) do { if constexpr (STAMPS) { if (p.dbg_cycles && (cond)) p.dbg_cycles[(size_t)g * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
    PHASE_STAMP(24, threadIdx.x == 0);
    // wave 0: the game's search, wave 1: its helper (encodes the leaf's planes while wave 0 generates the moves); both run
    // inside the tower's prologue (tower_body, Pre), after every wave has requested its first weights
    if (threadIdx.x == 0) s_box.state = 0;   // LDS starts with whatever the previous workgroup left: clear the mailbox ...
    __syncthreads();                         // ... before wave 1 can poll it or wave 0 post to it
    auto search = [&](int wave) -> bool {
        if (wave == 1) {
            sc::dev_encode_helper(p, g, lane, &s_box, s_stage, &s_pos, s_ps, s_hist);
            return true;
        }
        constexpr bool SC_ST = STAMPS || SC_ST_DEFAULT;
        SC_STAMP(0);
        sc::GameCtl cs_pre{};
        bool cs_pre_valid = false;
        if (do_expand) {
            sc::dev_expand<SC_ST>(p, g, lane, &s_pos, cs_pre, cs_pre_valid);
            __builtin_amdgcn_wave_barrier();
        }
        SC_STAMP(1);
        const bool need_net = sc::dev_select<false, SC_ST>(p, g, lane, s_stage, s_moves, &s_pos, s_ps, s_hist, cs_pre, cs_pre_valid, &s_box);
        PHASE_STAMP(25, lane == 0);
        return need_net;
    };
    scnn::Fc1Hand fh;
    const bool ran = scnn::tower_body<P, C, RS, TPI, AB>(A, g, s_stage, search, fh);
    PHASE_STAMP(26, threadIdx.x == 0);
    if (A.fc1_arrive) fc1_tail(A, p, g, ran, fh);
    PHASE_STAMP(27, threadIdx.x == 0);
#undef PHASE_STAMP
}

}  // namespace scstep

namespace scl {
#define K_STEP(P, C, RS, TPI, AB) scstep::k_step<scnn::P, C, RS, TPI, AB>
#define K_STEP_STAMPED scstep::k_step<scnn::PrecBF16, 128, SC_T32_RS, SC_T32_TPI, SC_T32_AB, true>
const char* step_init() {
    const void* kn[4] = {reinterpret_cast<const void*>(&K_STEP(PrecBF16, 128, SC_T32_RS, SC_T32_TPI, SC_T32_AB)),
                         reinterpret_cast<const void*>(&K_STEP(PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32W_AB)),
                         reinterpret_cast<const void*>(&K_STEP(PrecFP8, 128, SC_T8_RS, SC_T8_TPI, SC_T8_AB)),
                         reinterpret_cast<const void*>(&K_STEP(PrecFP8, 256, SC_T8W_RS, SC_T8W_TPI, SC_T8W_AB))};
    for (int i = 0; i < 4; i++) {
        hipError_t e = hipFuncSetAttribute(kn[i], hipFuncAttributeMaxDynamicSharedMemorySize, scnn::tower32_lds_bytes(i & 1 ? 256 : 128, i >= 2 && i < 4));
        if (e != hipSuccess) return hipGetErrorString(e);
    }
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K_STEP_STAMPED), hipFuncAttributeMaxDynamicSharedMemorySize, scnn::tower32_lds_bytes(128));
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
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, K_STEP(PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32W_AB), 256, scnn::tower32_lds_bytes(256));
    return e == hipSuccess && n > 0 ? n : 1;
}
void step(const scnn::TowerArgs& a, const sc::SpParams& p, int do_expand, hipStream_t s) {
    const dim3 grid(p.n_slots), block(256);
    if (a.net.fp8 && a.net.C == 128)
        hipLaunchKernelGGL((K_STEP(PrecFP8, 128, SC_T8_RS, SC_T8_TPI, SC_T8_AB)), grid, block, scnn::tower32_lds_bytes(128, true), s, a, p, do_expand);
    else if (a.net.fp8)
        hipLaunchKernelGGL((K_STEP(PrecFP8, 256, SC_T8W_RS, SC_T8W_TPI, SC_T8W_AB)), grid, block, scnn::tower32_lds_bytes(256, true), s, a, p, do_expand);
    else if (a.net.C == 128 && p.dbg_cycles)   // stamps switched on (sc_selfplay_debug_cycles): the stamped instantiation
        hipLaunchKernelGGL((K_STEP_STAMPED), grid, block, scnn::tower32_lds_bytes(128), s, a, p, do_expand);
    else if (a.net.C == 128)
        hipLaunchKernelGGL((K_STEP(PrecBF16, 128, SC_T32_RS, SC_T32_TPI, SC_T32_AB)), grid, block, scnn::tower32_lds_bytes(128), s, a, p, do_expand);
    else
        hipLaunchKernelGGL((K_STEP(PrecBF16, 256, SC_T32W_RS, SC_T32W_TPI, SC_T32W_AB)), grid, block, scnn::tower32_lds_bytes(256), s, a, p, do_expand);
}
}  // namespace scl
