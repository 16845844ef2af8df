// mcts_kernels.hpp -- PUCT search on the GPU: one 64-lane wavefront per concurrent game.
//
// Rewrites src/mcts.rs (Node/Cursor/uct/find_max/backward/select/mcts/step), the Game::predict
// front half (legal moves, terminal test, _encode: src/backends/torch.rs:89-113, src/chess.rs:845-877)
// and the per-ply driver of src/main.rs:168-233 as HIP kernels.
//
// Data layout in HBM (per game slot g; all arrays are contiguous per slot so one wave's accesses
// coalesce): SoA node pool N/W/P/U/MV/NC/FC/PS[g*node_cap + i] with the children of a node stored
// contiguously in python-chess move order (lane = child in PUCT), Position records for the game
// line (hist) and for expanded nodes (tpos), the last path, and the NN input/outputs of the leaf.
// Priors are cached at expansion (the reference re-evaluates the net at every node of every
// descent, src/mcts.rs:152; the net is deterministic so the search is identical).
//
// Scalar chess logic (make_move, move generation) is executed wave-uniformly (all lanes compute
// the same values: no divergence, no broadcasts); PUCT argmax, repetition scan, plane encoding,
// child initialisation and backup are lane-parallel with wave shuffles/ballots.
//
// This translation unit is compiled with -ffp-contract=off: the PUCT arithmetic must round
// exactly like the reference's f32 expression (src/mcts.rs:69-75), which Rust never contracts.
// SC_NO_KERNELS: only the device functions (step_kernels.hip reuses dev_expand / dev_select inside its own kernel).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "chess_history.hpp"
#include "chess_rules_wave.hpp"
#include "mcts_types.hpp"
#include "value_tail.hpp"

namespace sc {

// ------------------------------------------------------------------ wave helpers (64 lanes)
// Reductions inside the descent are latency chains (one per tree level): four DPP steps inside each row of 16 lanes
// (quad_perm, row_half_mirror, row_mirror: plain VALU moves) and four readlanes to the scalar unit, instead of six
// dependent trips through the LDS crossbar (ds_bpermute).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_sum_i(int v) {
    v += dpp_i<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_i<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_i<0x141>(v);   // row_half_mirror
    v += dpp_i<0x140>(v);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
// index of the maximal u over the lanes with idx >= 0; ties go to the LARGER index (Iterator::max_by keeps the last
// maximum, src/mcts.rs:78-88).  u must be finite.  Returns -1 when no lane has a candidate.  Wave-uniform result.
__device__ __forceinline__ int wave_argmax_last(float u, int idx) {
    // order-preserving map of a finite float to unsigned (+0.0 added first: -0.0 and +0.0 compare equal as floats)
    unsigned b = __builtin_bit_cast(unsigned, u + 0.0f);
    unsigned key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    int hi = idx >= 0 ? (int)key : 0, lo = idx;   // compared as unsigned; lo = -1 marks "none" (never wins: hi = 0 ... see below)
    auto step = [&](int ohi, int olo) {
        const bool take = olo >= 0 && (lo < 0 || (unsigned)ohi > (unsigned)hi || ((unsigned)ohi == (unsigned)hi && olo > lo));
        hi = take ? ohi : hi;
        lo = take ? olo : lo;
    };
    step(dpp_i<0xB1>(hi), dpp_i<0xB1>(lo));
    step(dpp_i<0x4E>(hi), dpp_i<0x4E>(lo));
    step(dpp_i<0x141>(hi), dpp_i<0x141>(lo));
    step(dpp_i<0x140>(hi), dpp_i<0x140>(lo));
    int bh = __builtin_amdgcn_readlane(hi, 0), bl = __builtin_amdgcn_readlane(lo, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const int oh = __builtin_amdgcn_readlane(hi, r), ol = __builtin_amdgcn_readlane(lo, r);
        const bool take = ol >= 0 && (bl < 0 || (unsigned)oh > (unsigned)bh || ((unsigned)oh == (unsigned)bh && ol > bl));
        bh = take ? oh : bh;
        bl = take ? ol : bl;
    }
    return bl;
}
// Same contract for the common case of ONE candidate per lane whose index is its lane number (nodes with <= 64
// children): the wave maximum of the keys (4 DPP max steps), then the HIGHEST lane holding it (ballot + find-last-set)
// -- a third of the instructions of the (key, index) pair reduction above, on every level of every descent.
__device__ __forceinline__ int wave_argmax_last_lane(float u, bool has) {
    unsigned b = __builtin_bit_cast(unsigned, u + 0.0f);
    unsigned key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // >= 0x00800000 for every finite float
    int k = has ? (int)key : 0;
    auto umax = [](int a, int b2) { return (int)((unsigned)a > (unsigned)b2 ? (unsigned)a : (unsigned)b2); };
    int m = k;
    m = umax(m, dpp_i<0xB1>(m));
    m = umax(m, dpp_i<0x4E>(m));
    m = umax(m, dpp_i<0x141>(m));
    m = umax(m, dpp_i<0x140>(m));
    const int wm = umax(umax(__builtin_amdgcn_readlane(m, 0), __builtin_amdgcn_readlane(m, 16)),
                        umax(__builtin_amdgcn_readlane(m, 32), __builtin_amdgcn_readlane(m, 48)));
    const unsigned long long mask = __ballot(has && k == wm);
    return mask ? 63 - __clzll((long long)mask) : -1;
}
// float sum over the wave in the fixed order the value tail uses (value_tail.hpp)
__device__ __forceinline__ float wave_sum_f_dpp(float v) { return scvt::wave_sum_fixed(v); }
__device__ inline float wave_sum_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Synchronisation inside the search functions (dev_expand, dev_select, finish_game): each game is searched by ONE
// wavefront -- the whole workgroup of k_mcts, or wave 0 of the tower's workgroup in the fused step kernel (k_step), where a
// workgroup barrier would wait for waves that never come.  A wave's LDS and vector-memory operations take effect in
// program order; what is needed between a store by one lane and a load by another is that the compiler keeps that order
// and the operations have completed: a workgroup-scope fence (s_waitcnt) plus a wave barrier (scheduling only).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Promote a wave-uniform value to SGPRs.  All lanes of a game's wave run the scalar chess logic on identical
// data; telling the compiler so (readfirstlane) moves that logic -- 64-bit bitboard arithmetic, bit scans, bit
// reversal, loop control -- from the vector ALU (2 x 32-bit ops, exec-mask branches) onto the scalar unit.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bb_t uniform(bb_t v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((bb_t)hi << 32) | lo;
}
__device__ __forceinline__ Position uniform(const Position& q) {
    Position r;
#pragma unroll
    for (int t = 0; t < 6; t++) r.pcs[t] = uniform(q.pcs[t]);
    r.occ[0] = uniform(q.occ[0]);
    r.occ[1] = uniform(q.occ[1]);
    r.key = uniform(q.key);
    r.turn = (uint8_t)uniform((int)q.turn);
    r.castling = (uint8_t)uniform((int)q.castling);
    r.ep = (int8_t)uniform((int)q.ep);
    r.flags = (uint8_t)uniform((int)q.flags);
    r.halfmove = (uint16_t)uniform((int)q.halfmove);
    r.fullmove = (uint16_t)uniform((int)q.fullmove);
    return r;
}
__device__ __forceinline__ NodeHdr uniform(const NodeHdr& q) {
    NodeHdr r;
    r.fc = uniform(q.fc);
    r.nc = (uint16_t)uniform((int)q.nc);
    r.ps = (uint16_t)uniform((int)q.ps);
    return r;
}
__device__ __forceinline__ GameCtl uniform(const GameCtl& q) {
    GameCtl r;
    r.status = uniform(q.status);
    r.ply = uniform(q.ply);
    r.sim = uniform(q.sim);
    r.n_nodes = uniform(q.n_nodes);
    r.n_exp = uniform(q.n_exp);
    r.leaf = uniform(q.leaf);
    r.path_len = uniform(q.path_len);
    r.leaf_kind = uniform(q.leaf_kind);
    r.n_legal = uniform(q.n_legal);
    r.leaf_value = __builtin_bit_cast(float, uniform(__builtin_bit_cast(int, q.leaf_value)));
    r.err = (uint32_t)uniform((int)q.err);
    r.trace_slot = uniform(q.trace_slot);
    r.game_id = uniform((bb_t)q.game_id);
    r.start_ply = uniform(q.start_ply);
    r.rollout_cur = uniform(q.rollout_cur);
    return r;
}

// chain of positions: game history, then the tree path, then the leaf being created
struct DevChain {
    const Position* hist;
    int root_ply;
    const Position* tpos;
    const uint16_t* ps_by_depth;  // LDS: tpos slot of the path node at depth d (expanded nodes only)
    const Position* leaf;
    int leaf_idx;
    __device__ const Position& pos(int i) const {
        if (i <= root_ply) return hist[i];
        if (i == leaf_idx) return *leaf;
        return tpos[ps_by_depth[i - root_ply]];
    }
};
struct HistChain {
    const Position* hist;
    __device__ const Position& pos(int i) const { return hist[i]; }
};

// is_repetition(2) / is_repetition(3) of the position at chain index idx, lane-parallel.
// Lane L looks at i = idx-L: the walk of python-chess is_repetition stops at the first i whose
// incoming move was irreversible, and compares pos(i-1) otherwise.
template <class Chain>
__device__ inline uint8_t rep_flags_wave(const Chain& ch, int idx, bb_t key0, int lane) {
    int matches = 0;
    for (int base = 0;; base += 64) {
        int i = idx - base - lane;
        bool valid = i >= 1;
        bool irrev = false, match = false;
        if (valid) {
            irrev = (ch.pos(i).flags & F_IRREV) != 0;
            match = ch.pos(i - 1).key == key0;
        }
        unsigned long long stopmask = __ballot(irrev || !valid);
        unsigned long long matchmask = __ballot(match && valid);
        bool stopped = stopmask != 0;
        if (stopped) {
            int first = __ffsll((long long)stopmask) - 1;
            matchmask &= first == 0 ? 0ULL : (~0ULL >> (64 - first));
        }
        matches += __popcll(matchmask);
        if (stopped || matches >= 2) break;
    }
    return (uint8_t)((matches >= 1 ? F_REP2 : 0) | (matches >= 2 ? F_REP3 : 0));
}

// Stage the <=8 positions _encode looks at (idx, idx-1, ...) into LDS with two dependent round trips in total:
// lane l fetches 8-byte word (l & 7) [and word 8/9 for l&7 < 2] of history entry l >> 3.
template <class Chain>
__device__ inline void stage_history(const Chain& ch, int idx, int lane, Position* s_hist) {
    const int j = lane >> 3, w = lane & 7;
    if (j <= idx) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&ch.pos(idx - j));
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&s_hist[j]);
        dst[w] = src[w];
        if (w < 2) dst[8 + w] = src[8 + w];
    }
}

// _encode (src/chess.rs:845-877) for the position at history depth 0 of s_hist (newest first): lane = output
// pixel.  stage: 7168 B of LDS; out: int8[64][112] in HBM.
__device__ inline void encode_wave(const Position* s_hist, int n_hist, int lane, int8_t* stage, int8_t* out, int32_t* meta_out) {
    uint4* cell16 = reinterpret_cast<uint4*>(stage + lane * 112);
#pragma unroll
    for (int k = 0; k < 7; k++) cell16[k] = make_uint4(0, 0, 0, 0);
    int8_t* cell = stage + lane * 112;
    const int turn = s_hist[0].turn;
    int src = turn == BLACK ? (lane ^ 56) : lane;
    bb_t sb = bit(src);
    for (int j = 0; j < n_hist; j++) {
        const Position& h = s_hist[j];
        bb_t ow = h.occ[WHITE], ob = h.occ[BLACK];
        if ((ow | ob) & sb) {
            int t = 0;
#pragma unroll
            for (int k = 1; k < 6; k++)
                if (h.pcs[k] & sb) t = k;
            int is_white = (ow & sb) ? 1 : 0;
            int mover_side = turn == BLACK ? !is_white : is_white;
            cell[14 * j + t + (mover_side ? 0 : 6)] = 1;
        }
        uint8_t f = h.flags;
        cell[14 * j + 12] = (f & F_REP2) ? 1 : 0;
        cell[14 * j + 13] = (f & F_REP3) ? 1 : 0;
    }
    if (out) {   // (the fused step kernel hands the planes to the network in LDS: no copy to HBM)
        uint4* o16 = reinterpret_cast<uint4*>(out + lane * 112);
#pragma unroll
        for (int k = 0; k < 7; k++) o16[k] = cell16[k];
    }
    if (lane == 0) {
        int32_t m[7];
        encode_meta(s_hist[0], m);
#pragma unroll
        for (int k = 0; k < 7; k++) meta_out[k] = m[k];
        meta_out[7] = 0;
    }
}

// ------------------------------------------------------------------ Dirichlet(0.3) root noise
// get_noise (src/mcts.rs:123-130).  The reference draws from thread_rng; here a counter-based
// stream keyed by (seed, game, ply, sim, child) -- parity is distributional only.
__device__ inline float u01_open(uint64_t& st) {
#pragma clang fp contract(off)   // exact f32 like the reference, in whichever translation unit this is compiled (see wave_sync)
    st = mix64(st);
    return ((float)(st >> 40) + 0.5f) * (1.0f / 16777216.0f);
}
// Gamma(0.3, 1) sample for the Dirichlet(0.3) root noise (src/mcts.rs:123-130): Gamma(1.3) by Marsaglia-Tsang times
// U^(1/0.3).  The draw happens for every child at EVERY simulation, on the critical path of the descent, so it uses
// the hardware transcendentals (v_log / v_exp / v_cos / v_sqrt, ~1 ulp) instead of the correctly rounded library
// routines (10x the instructions): the reference's noise comes from thread_rng, parity is distributional
// (tests: test_root_noise_is_dirichlet).
__device__ inline float gamma03(uint64_t st) {
#pragma clang fp contract(off)   // exact f32 like the reference, in whichever translation unit this is compiled (see wave_sync)
    // raw hardware transcendentals only (v_log_f32 = log2, v_exp_f32 = 2^x): the library's natural-log / exp wrappers expand
    // differently with and without FMA contraction, and this function is compiled in two translation units
    // (mcts_kernels.hip, step_kernels.hip) that must draw the same noise
    const float LN2 = 0.69314718f;
    const float alpha = 0.3f;
    const float boost = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u01_open(st)) * (1.0f / alpha));   // u^(1/alpha)
    const float d = alpha + 1.0f - 1.0f / 3.0f;
    const float c = 0.3390317518f;  // 1 / sqrt(9 d)
    for (int it = 0; it < 64; it++) {
        float a = u01_open(st), b = u01_open(st);
        float x = __builtin_amdgcn_sqrtf(-2.0f * LN2 * __builtin_amdgcn_logf(a)) * __builtin_amdgcn_cosf(b);   // v_cos_f32 takes revolutions
        float v = 1.0f + c * x;
        if (v <= 0.0f) continue;
        v = v * v * v;
        float u = u01_open(st);
        if (LN2 * __builtin_amdgcn_logf(u) < 0.5f * x * x + d - d * v + d * (LN2 * __builtin_amdgcn_logf(v))) return boost * d * v;
    }
    return boost * d;
}

// ------------------------------------------------------------------ select (src/mcts.rs:132-227)
constexpr int DEPTH_LDS = 1024;  // path entries tracked in LDS (a deeper path sets ERR_DEPTH_OVERFLOW)

#ifdef SC_EXP   // experiment builds: stamps inside the expansion (slots 16..), tools/dbg_expand.py
#define SC_XSTAMP(k) SC_STAMP(k)
#else
#define SC_XSTAMP(k)
#endif
// Cycle stamps of the search's phases (tools/dbg_cycles.py).  Compiled into the kernel that runs every step they cost 0.7 % of
// the headline even while switched off (same-box A/B, tools/ab_r02.py): SC_ST is a template argument of the functions that carry
// them -- true in k_mcts and in the stamped instantiation of the fused step kernel, which the engine launches only while stamps are
// switched on; experiment builds stamp everywhere.
#ifdef SC_EXP
#define SC_ST_DEFAULT true
#else
#define SC_ST_DEFAULT false
#endif
#define SC_STAMP(k)                                                                       \
    do {                                                                                  \
        if constexpr (SC_ST) {                                                            \
            if (p.dbg_cycles && lane == 0) p.dbg_cycles[(size_t)g * 32 + (k)] = clock64(); \
        }                                                                                 \
    } while (0)

// Hand-off to a helper wavefront (fused step kernel): once the leaf position and its repetition flags stand, the plane
// encoding (history loads + 7 KB of LDS writes) is independent of move generation; wave 1 of the workgroup, idle during
// the search, does it while this wave generates the moves.  An LDS mailbox: the searching wave writes idx / root_ply,
// then state (1 = encode, 2 = nothing to do, 3 = the leaf's repetition flags first -- answered with state 4 and the flags in
// `pad` --, then encode); the helper polls state.
struct HelperBox {
    int state, idx, root_ply, pad;
};
__device__ __forceinline__ void helper_post(HelperBox* box, int lane, int state, int idx, int root_ply) {
    if (!box) return;
    if (lane == 0) {
        box->idx = idx;
        box->root_ply = root_ply;
    }
    wave_sync();   // the payload (and s_leaf / s_ps before it) has landed in LDS before the state word is written
    if (lane == 0) *reinterpret_cast<volatile int*>(&box->state) = state;
}
// the helper wave: waits for the mailbox, encodes the planes of the leaf into `stage` (and its meta row)
__device__ __forceinline__ void dev_encode_helper(const SpParams& p, int g, int lane, HelperBox* box, int8_t* s_stage, Position* s_leaf_p,
                                                  uint16_t* s_ps, Position* s_hist) {
    int st = 0;
    for (int spin = 0; spin < (1 << 22) && st == 0; spin++) {   // bounded: a wave never hangs on a missing post
        st = *reinterpret_cast<volatile int*>(&box->state);
        if (st == 0) __builtin_amdgcn_s_sleep(2);
    }
    st = __builtin_amdgcn_readfirstlane(st);
    if (st == 0 && lane == 0) atomicOr(&p.cnt->err, ERR_HELPER_TIMEOUT);   // never seen: the search wave posts on every path
    if (st != 1 && st != 3) return;
    wave_sync();
    const int idx = __builtin_amdgcn_readfirstlane(box->idx), root_ply = __builtin_amdgcn_readfirstlane(box->root_ply);
    DevChain ch{p.hist + (size_t)g * p.hist_cap, root_ply, p.tpos + (size_t)g * p.tpos_cap, s_ps, s_leaf_p, idx};
    if (st == 3) {
        // the leaf's repetition flags (move generation does not need them: the search wave is already generating the moves
        // and collects the flags from the mailbox afterwards)
        const bb_t key0 = s_leaf_p->key;
        const uint8_t rf = (uint8_t)__builtin_amdgcn_readfirstlane((int)rep_flags_wave(ch, idx, key0, lane));
        const uint8_t fl = (uint8_t)((s_leaf_p->flags & F_IRREV) | rf);
        wave_sync();   // every lane has read the old flags
        if (lane == 0) {
            s_leaf_p->flags = fl;
            box->pad = fl;
        }
        wave_sync();
        if (lane == 0) *reinterpret_cast<volatile int*>(&box->state) = 4;
    }
    stage_history(ch, idx, lane, s_hist);
    wave_sync();
    encode_wave(s_hist, idx < 7 ? idx + 1 : 8, lane, s_stage, nullptr, p.meta + (size_t)g * 8);
}

// Returns true when the selected leaf needs a network evaluation (planes, legal moves and action indices are then in
// place).  PLANES_TO_HBM = false: the planes stay in s_stage (fused step kernel).  box != nullptr: the planes are encoded
// by the helper wave (above) instead of this one.
template <bool PLANES_TO_HBM = true, bool SC_ST = true>
__device__ __forceinline__ bool dev_select(const SpParams& p, int g, int lane, int8_t* s_stage, move_t* s_moves, Position* s_leaf_p,
                                        uint16_t* s_ps, Position* s_hist, const GameCtl& cs_pre, bool cs_pre_valid,
                                        HelperBox* box = nullptr) {
#pragma clang fp contract(off)
    Position& s_leaf = *s_leaf_p;
    SC_STAMP(2);
    GameCtl& c = p.ctl[g];
    GameCtl cs = cs_pre;
    if (!cs_pre_valid) cs = uniform(c);  // one 64-byte fetch instead of a chain of dependent field loads
    // The root position, the root header and the root's children ride in ONE round trip (their addresses depend on g
    // only: the tree is rebuilt every ply with the root at node 0 and its children at nodes 1..nc, first expansion).
    // As in dev_expand: all loads first, unguarded (clamped index; lanes past the child count are masked where the
    // values are used), and the wave-uniform ones move to SGPRs only after the last load has been issued.
    const Position root_raw = p.tpos[(size_t)g * p.tpos_cap];
    const NodeHdr hdr_raw = p.H[(size_t)g * p.node_cap];
    const size_t nb0 = (size_t)g * p.node_cap + (1 + lane < p.node_cap ? 1 + lane : p.node_cap - 1);
    const int pf_n = p.N[nb0];
    const float pf_w = p.W[nb0], pf_p = p.P[nb0];
    const NodeHdr pf_h = p.H[nb0];
    __builtin_amdgcn_sched_barrier(0);
    const Position root = uniform(root_raw);
    NodeHdr hdr = uniform(hdr_raw);
    if (cs.status != ST_ACTIVE) {
        if (lane == 0) c.leaf_kind = LK_NONE;
        helper_post(box, lane, 2, 0, 0);
        return false;
    }
    const size_t nb = (size_t)g * p.node_cap;
    const int32_t* N = p.N + nb;
    const float* W = p.W + nb;
    const float* P = p.P + nb;
    float* U = p.U + nb;
    const uint16_t* MV = p.MV + nb;
    const NodeHdr* H = p.H + nb;
    int32_t* path = p.path + (size_t)g * p.max_depth;
    const Position* hist = p.hist + (size_t)g * p.hist_cap;
    Position* tpos = p.tpos + (size_t)g * p.tpos_cap;
    const int root_ply = cs.ply;
    const int root_turn = root.turn;
    const int dmax = p.max_depth < DEPTH_LDS ? p.max_depth : DEPTH_LDS;

    int node = 0, depth = 0, parent_ps = 0;
    if (lane == 0) {
        path[0] = 0;
        s_ps[0] = 0;
    }
    uint32_t err = 0;
    for (;;) {
        const int nc = hdr.nc;
        if (nc == 0) break;
        const int fc = hdr.fc;
        // One level of the descent, generic in the number of 64-child rounds it is compiled for: nodes with more than
        // 64 children are rare (the common case is ONE round), and the 4-round code carries four sets of statistics,
        // guards and selects through the PUCT arithmetic of every level.
        const int nr = (nc + 63) >> 6;  // rounds of 64 children (wave-uniform): usually 1
        int best_i = 0;
        NodeHdr nxt;
        auto level = [&](auto nrc) {
#pragma clang fp contract(off)
            constexpr int NRM = decltype(nrc)::value;
            // children statistics AND their headers in one round trip (lane owns children lane, lane+64, ...)
            int cn[NRM];
            float cw[NRM], cp[NRM];
            NodeHdr ch_[NRM];
#pragma unroll
            for (int r = 0; r < NRM; r++) {
                cn[r] = 0;
                cw[r] = 0.f;
                cp[r] = 0.f;
                ch_[r] = NodeHdr{-1, 0, 0};
                if (r < nr) {
                    int i = lane + 64 * r;
                    bool ok = i < nc;
                    if (r == 0 && depth == 0 && fc == 1) {   // prefetched with the control block
                        cn[0] = ok ? pf_n : 0;
                        cw[0] = ok ? pf_w : 0.f;
                        cp[0] = ok ? pf_p : 0.f;
                        ch_[0] = ok ? pf_h : NodeHdr{-1, 0, 0};
                    } else {
                        cn[r] = ok ? N[fc + i] : 0;
                        cw[r] = ok ? W[fc + i] : 0.f;
                        cp[r] = ok ? P[fc + i] : 0.f;
                        ch_[r] = ok ? H[fc + i] : NodeHdr{-1, 0, 0};
                    }
                }
            }
            if (nc > 1) {
                // side to move at `node`: root_turn flipped per depth; reverse_q = Black to move (torch.rs:49-52)
                const bool reverse_q = ((root_turn ^ (depth & 1)) == BLACK);
                const bool noisy = depth == 0 && p.with_noise;
                float* nz = p.noise + (size_t)g * MAXC;
                float nzv[NRM];
#pragma unroll
                for (int r = 0; r < NRM; r++) nzv[r] = 0.f;
                if (noisy) {
                    if (!p.external_noise) {
                        float gsum = 0.0f;
#pragma unroll
                        for (int r = 0; r < NRM; r++) {
                            int i = lane + 64 * r;
                            if (i < nc) {
                                nzv[r] = gamma03(sc_rng(p.seed, cs.game_id, (uint64_t)root_ply, 3, (uint64_t)cs.sim * 256 + (uint64_t)i));
                                gsum += nzv[r];
                            }
                        }
                        gsum = wave_sum_f_dpp(gsum);
#pragma unroll
                        for (int r = 0; r < NRM; r++) {
                            int i = lane + 64 * r;
                            nzv[r] = nzv[r] / gsum;
                            if (i < nc) nz[i] = nzv[r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < NRM; r++) {
                            int i = lane + 64 * r;
                            if (i < nc) nzv[r] = nz[i];
                        }
                    }
                }
                int tot = 0;
#pragma unroll
                for (int r = 0; r < NRM; r++) tot += cn[r];
                tot = wave_sum_i(tot);
                const float sqrt_total = sqrtf((float)tot);
                float best_u = 0.0f;
                best_i = -1;
#pragma unroll
                for (int r = 0; r < NRM; r++) {
                    int i = lane + 64 * r;
                    if (r < nr && i < nc) {
                        float prior = cp[r];
                        if (noisy) prior = prior * (1.0f - p.epsilon) + nzv[r] * p.epsilon;  // mcts.rs:181
                        // uct(): src/mcts.rs:69-75
                        float average_award = cw[r] / ((float)cn[r] + 1e-4f) * (reverse_q ? -1.0f : 1.0f);
                        float exploration = (sqrt_total + 0.01f) / (1.0f + (float)cn[r]) * p.cpuct * prior;
                        float u = average_award + exploration;
                        U[fc + i] = u;
                        if (!isfinite(u)) err |= ERR_NONFINITE_UCT;
                        if (best_i < 0 || u >= best_u) {  // later index wins ties (max_by keeps the last maximum)
                            best_u = u;
                            best_i = i;
                        }
                    }
                }
                if constexpr (NRM == 1) best_i = wave_argmax_last_lane(best_u, best_i >= 0);   // best_i is the lane number here
                else best_i = wave_argmax_last(best_u, best_i);
            }
            best_i = __builtin_amdgcn_readfirstlane(best_i);
            // header of the chosen child: owned by lane best_i & 63, register best_i >> 6
            const int rr = best_i >> 6;
            NodeHdr mine = ch_[0];
#pragma unroll
            for (int r = 1; r < NRM; r++)
                if (rr == r) mine = ch_[r];
            nxt.fc = __builtin_amdgcn_readlane(mine.fc, best_i & 63);
            int packed = __builtin_amdgcn_readlane((int)mine.nc | ((int)mine.ps << 16), best_i & 63);
            nxt.nc = (uint16_t)(packed & 0xffff);
            nxt.ps = (uint16_t)((unsigned)packed >> 16);
        };
        if (nr == 1) level(std::integral_constant<int, 1>{});
        else level(std::integral_constant<int, 4>{});
        parent_ps = hdr.ps;
        node = fc + best_i;
        hdr = nxt;
        depth++;
        if (depth >= dmax) {
            err |= ERR_DEPTH_OVERFLOW;
            depth--;
            break;
        }
        if (lane == 0) {
            path[depth] = node;
            s_ps[depth] = hdr.ps;
        }
    }
    SC_STAMP(3);
    if (p.dbg_cycles && lane == 0) p.dbg_cycles[(size_t)g * 32 + 7] = depth;   // developer stamp: levels walked
    unsigned long long anyerr = __ballot(err != 0);
    if (anyerr) {
        for (int o = 32; o > 0; o >>= 1) err |= __shfl_xor(err, o, 64);
        if (lane == 0) {
            c.err = cs.err | err;
            atomicOr(&p.cnt->err, (int)err);
        }
    }
    const int fcl = hdr.fc;
    if (lane == 0) {
        c.leaf = node;
        c.path_len = depth + 1;
    }
    if (fcl <= -2) {  // terminal seen before: predict() returns the same outcome again (torch.rs:98-106)
        if (lane == 0) {
            c.leaf_kind = LK_TERM_CACHED;
            c.leaf_value = fcl == -2 ? 0.0f : fcl == -3 ? 1.0f : -1.0f;
            c.n_legal = 0;
        }
        helper_post(box, lane, 2, 0, 0);
        return false;
    }
    // position of the leaf (wave-uniform)
    Position pos;
    if (depth == 0) {
        pos = root;
    } else {
        pos = uniform(tpos[parent_ps]);
        make_move(pos, (move_t)__builtin_amdgcn_readfirstlane((int)MV[node]));  // state.advance (mcts.rs:224)
    }
    if (lane == 0) s_leaf = pos;
    wave_sync();  // s_leaf, s_ps visible
    DevChain ch{hist, root_ply, tpos, s_ps, &s_leaf, root_ply + depth};
    const bool rep_by_helper = box && depth > 0;   // the helper wave scans for repetitions too (a round trip off this wave's chain)
    if (depth > 0 && !box) {
        uint8_t rf = (uint8_t)__builtin_amdgcn_readfirstlane((int)rep_flags_wave(ch, root_ply + depth, pos.key, lane));
        pos.flags = (uint8_t)((pos.flags & F_IRREV) | rf);
        wave_sync();
        if (lane == 0) s_leaf.flags = pos.flags;
        wave_sync();
    }
    SC_STAMP(4);
    // history for the encoder: issued now so the loads overlap move generation -- or the whole encoding handed to the
    // helper wave (a terminal leaf wastes its work: nothing reads the planes then)
    if (box) helper_post(box, lane, rep_by_helper ? 3 : 1, root_ply + depth, root_ply);
    else stage_history(ch, root_ply + depth, lane, s_hist);
    // scratch slot for the expansion (claimed in dev_expand if the leaf is not terminal); with the helper's flags: below
    if (lane == 0 && !rep_by_helper) tpos[cs.n_exp] = pos;
    int n = 0;
    bool in_check = gen_legal_wave(pos, s_moves, lane, n);   // lane = square (chess_rules_wave.hpp)
    wave_sync();
    SC_STAMP(5);
    // --rollout-factor (src/main.rs:175-176): the ply's budget follows from the root's legal-move count, known here at
    // the first simulation of the ply (the only one whose leaf is the root)
    if (depth == 0 && p.rollout_factor > 0.f && lane == 0) {
        const int r = (int)((float)n * p.rollout_factor);
        c.rollout_cur = r < 300 ? r : 300;
    }
    if (n == 0) {
        if (lane == 0) {
            c.leaf_kind = LK_TERM_NEW;
            // winner -> +1 white / -1 black / 0 (torch.rs:100-104); checkmated side is the one to move
            c.leaf_value = in_check ? (pos.turn == WHITE ? -1.0f : 1.0f) : 0.0f;
            c.n_legal = 0;
        }
        return false;
    }
    uint16_t* lm = p.legal_mv + (size_t)g * MAXC;
    uint16_t* li = p.legal_idx + (size_t)g * MAXC;
    uint32_t bad = 0;
    for (int i = lane; i < n; i += 64) {
        move_t m = s_moves[i];
        int idx = move_index(m, pos.turn);
        if (idx < 0) {
            bad = 1;
            idx = 0;
        }
        lm[i] = m;
        li[i] = (uint16_t)idx;
    }
    if (__ballot(bad) && lane == 0) {
        c.err = cs.err | err | ERR_BAD_MOVE_INDEX;
        atomicOr(&p.cnt->err, ERR_BAD_MOVE_INDEX);
    }
    if (rep_by_helper) {
        int st = 3;
        for (int spin = 0; spin < (1 << 22) && st != 4; spin++) {   // (long answered: the scan is shorter than move generation)
            st = *reinterpret_cast<volatile int*>(&box->state);
            if (st != 4) __builtin_amdgcn_s_sleep(1);
        }
        st = __builtin_amdgcn_readfirstlane(st);
        wave_sync();
        if (st != 4 && lane == 0) atomicOr(&p.cnt->err, ERR_HELPER_TIMEOUT);
        pos.flags = (uint8_t)__builtin_amdgcn_readfirstlane(box->pad);
        if (lane == 0) tpos[cs.n_exp] = pos;
    }
    const int idx = root_ply + depth;
    if (!box) encode_wave(s_hist, idx < 7 ? idx + 1 : 8, lane, s_stage, PLANES_TO_HBM ? p.boards + (size_t)g * 7168 : nullptr, p.meta + (size_t)g * 8);
    if (lane == 0) {
        c.leaf_kind = LK_EVAL;
        c.n_legal = n;
        p.n_legal[g] = n;
    }
    SC_STAMP(6);
    return true;
}

// ------------------------------------------------------------------ find_max on given values (test aid, sc_debug_find_max)
// The two argmax forms of the descent on caller-provided PUCT values: out[0] = one-round form (n <= 64, lane = child),
// out[1] = four-round (value, index) pair form (n <= 256, lane owns children lane, lane+64, ...), exactly as `level`
// above combines them.  Lets a test place exact ties, -0.0 / +0.0 pairs and maxima in any lane and round.
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_debug_find_max(const float* u, int n, int* out) {
    const int lane = threadIdx.x;
    if (n <= 64) {
        const int r = wave_argmax_last_lane(lane < n ? u[lane] : 0.f, lane < n);
        if (lane == 0) out[0] = r;
    } else if (lane == 0) {
        out[0] = -2;
    }
    float best_u = 0.f;
    int best_i = -1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = lane + 64 * r;
        if (i < n) {
            const float v = u[i];
            if (best_i < 0 || v >= best_u) {
                best_u = v;
                best_i = i;
            }
        }
    }
    const int r4 = wave_argmax_last(best_u, best_i);
    if (lane == 0) out[1] = r4;
}
#endif

// ------------------------------------------------------------------ synthetic evaluator (tests)
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_synth_eval(SpParams p) {
    const int g = blockIdx.x, lane = threadIdx.x;
    GameCtl& c = p.ctl[g];
    if (c.status != ST_ACTIVE || c.leaf_kind != LK_EVAL) return;
    const Position& pos = p.tpos[(size_t)g * p.tpos_cap + c.n_exp];
    uint64_t h = synth_pos_hash(pos) ^ p.synth_salt;
    int n = c.n_legal;
    const uint16_t* lm = p.legal_mv + (size_t)g * MAXC;
    if (p.evaluator == SYNTH_UNIFORM) {
        // tie tests: every sibling has the same prior and every leaf the value 0, so all unvisited children of a node tie
        // exactly and find_max's LAST-maximum rule (src/mcts.rs:78-88) decides every descent
        for (int i = lane; i < n; i += 64) p.prior[(size_t)g * MAXC + i] = 1.0f / (float)n;
        if (lane == 0) p.value[g] = 0.0f;
        return;
    }
    // SYNTH_COARSE: 2-bit weights and values from {-0.5, 0, 0, 0.5}: exact PUCT ties between SOME siblings, next to
    // non-zero value sums (the hash evaluator's 24-bit priors never collide)
    const bool coarse = p.evaluator == SYNTH_COARSE;
    unsigned long long sum = 0;
    for (int i = lane; i < n; i += 64) sum += coarse ? 1u + (synth_weight(h, lm[i]) >> 22) : synth_weight(h, lm[i]);
    sum = wave_sum_u64(sum);
    float fs = (float)sum;
    for (int i = lane; i < n; i += 64)
        p.prior[(size_t)g * MAXC + i] = (float)(coarse ? 1u + (synth_weight(h, lm[i]) >> 22) : synth_weight(h, lm[i])) / fs;
    if (lane == 0) {
        const float v = synth_value(h);
        p.value[g] = coarse ? (v < -0.5f ? -0.5f : v >= 0.5f ? 0.5f : 0.0f) : v;
    }
}
#endif

// ------------------------------------------------------------------ game (re)start
// Game ordinal k (0-based on this handle) takes trace-ring row k % trace_cap, strictly after game k - trace_cap: the row
// must hold THAT game, finished (and, with trace_hold, released by the host: sc_selfplay_poll).  A long game next to slots
// that cycle through short ones can be lapped -- writing its rows would corrupt both traces -- and several waiting games
// can map to the same row (k + cap, k + 2 cap, ...): they start one after the other.  A slot that cannot start parks its
// ordinal in game_id (ST_PENDING) and retries at every simulation step.
__device__ inline void try_start_game(SpParams& p, int g, int lane, unsigned long long k) {
    GameCtl& c = p.ctl[g];
    const int ts = (int)(k % (unsigned long long)p.trace_cap);
    const int st = p.thdr[ts].state;
    const unsigned long long prev_id = p.thdr[ts].game_id;
    bool ok;
    if (k < (unsigned long long)p.trace_cap) ok = st == TR_FREE;   // first use of the row
    else ok = prev_id == p.first_game_id + k - (unsigned long long)p.trace_cap && (st == TR_FREE || (st == TR_DONE && !p.trace_hold));
    if (!ok) {
        if (lane == 0) {
            c.status = ST_PENDING;
            c.leaf_kind = LK_NONE;
            c.game_id = p.first_game_id + k;
        }
        return;
    }
    const size_t nb = (size_t)g * p.node_cap;
    if (lane == 0) {
        Position s;
        set_startpos(s);
        s.key = position_key(s);
        p.hist[(size_t)g * p.hist_cap] = s;
        p.tpos[(size_t)g * p.tpos_cap] = s;
        p.N[nb] = 0;
        p.W[nb] = 0.0f;
        p.P[nb] = 0.0f;
        p.U[nb] = 0.0f;
        p.MV[nb] = 0;
        p.H[nb] = NodeHdr{-1, 0, 0};
        c.status = ST_ACTIVE;
        c.ply = 0;
        c.start_ply = 0;
        c.sim = 0;
        c.n_nodes = 1;
        c.n_exp = 1;
        c.leaf = 0;
        c.path_len = 0;
        c.leaf_kind = LK_NONE;
        c.n_legal = 0;
        c.err = 0;
        c.rollout_cur = p.rollout;
        c.game_id = p.first_game_id + k;
        c.trace_slot = ts;
        TraceHdr& th = p.thdr[ts];
        th.n_steps = 0;
        th.has_outcome = 0;
        th.termination = 0;
        th.winner = -1;
        th.game_id = c.game_id;
        th.state = TR_LIVE;
    }
}
__device__ inline void start_new_game(SpParams& p, int g, int lane) {
    GameCtl& c = p.ctl[g];
    unsigned long long k = 0;
    if (lane == 0) k = atomicAdd(&p.cnt->next_game, 1ULL);
    k = __shfl(k, 0, 64);
    if (k >= (unsigned long long)p.total_games) {
        if (lane == 0) {
            c.status = ST_IDLE;
            c.leaf_kind = LK_NONE;
        }
        return;
    }
    try_start_game(p, g, lane, k);
}
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_init_slots(SpParams p) {
    const int g = blockIdx.x, lane = threadIdx.x;
    uint4* b = reinterpret_cast<uint4*>(p.boards + (size_t)g * 7168);
    for (int i = lane; i < 448; i += 64) b[i] = make_uint4(0, 0, 0, 0);
    if (lane < 8) p.meta[(size_t)g * 8 + lane] = 0;
    if (lane == 0) p.n_legal[g] = 0;
    // slot g starts with game g (a deterministic slot <-> game map at start; later games are drawn from the counter as
    // slots free up); no game has finished yet, so nothing else touches the counter during this launch
    if (g == 0 && lane == 0)
        atomicAdd(&p.cnt->next_game, (unsigned long long)(p.n_slots < p.total_games ? p.n_slots : p.total_games));
    if (g >= p.total_games) {
        if (lane == 0) {
            p.ctl[g].status = ST_IDLE;
            p.ctl[g].leaf_kind = LK_NONE;
        }
        return;
    }
    try_start_game(p, g, lane, (unsigned long long)g);
}
#endif

__device__ inline void finish_game(SpParams& p, int g, int lane, int has_outcome, int term, int winner) {
    GameCtl& c = p.ctl[g];
    if (lane == 0) {
        TraceHdr& th = p.thdr[c.trace_slot];
        th.n_steps = c.ply - c.start_ply;
        th.has_outcome = has_outcome;
        th.termination = term;
        th.winner = winner;
        th.game_id = c.game_id;
        __threadfence();
        th.state = TR_DONE;
        atomicAdd(&p.cnt->games_finished, 1);
        c.status = ST_FINISHED;
    }
    wave_sync();
    start_new_game(p, g, lane);
}

// ------------------------------------------------------------------ expand + backward + mcts::step
// mcts.rs:267-288 (expand, backward), then when the rollout count is reached the per-ply part of
// src/main.rs:198-233: snapshot root/children into the trace, mcts::step (mcts.rs:292-328), outcome.
// value head tail for one position (nn_kernels.hpp k_value_finish, fused here so that the search step needs
// no separate launch): + meta columns + bias, ReLU, Linear 128->1, tanh, times (2*turn-1)
// Two halves: every address depends on the game slot only, so the loads are requested at the very top of the expansion,
// together with the control block (one round trip earlier than the path statistics, which need the control block).
using scvt::ValueTail;
__device__ __forceinline__ void value_tail_issue(const SpParams& p, int g, int lane, ValueTail& t) {
    const float* wf = p.vf_w;
    const int32_t* meta = p.meta + (size_t)g * 8;
#pragma unroll
    for (int k = 0; k < 7; k++) t.meta[k] = meta[k];
    // lane owns output columns 2*lane, 2*lane+1; ALL split-K partials are requested before the first add (one L2
    // round trip instead of one per 32 partials), then summed in fixed ascending order by value_tail.hpp's
    // value_tail_compute -- the function k_value_finish calls too (tests/test_gpu_netloop.py: bitwise identical)
    const int j = 2 * lane;
    const float* vp = p.vpart + (size_t)g * 128 + j;
    const size_t vstride = (size_t)p.n_slots * 128;
    // split-K is 32 or 64 (engine.hip): two unconditional batches -- a per-partial bound check makes the compiler
    // branch around (and wait for) every single load
#pragma unroll
    for (int ks = 0; ks < 32; ks++) t.acc[ks] = *reinterpret_cast<const float2*>(vp + (size_t)ks * vstride);
    if (p.vf_ksplit > 32) {
#pragma unroll
        for (int ks = 32; ks < 64; ks++) t.acc[ks] = *reinterpret_cast<const float2*>(vp + (size_t)ks * vstride);
    } else {
#pragma unroll
        for (int ks = 32; ks < 64; ks++) t.acc[ks] = make_float2(0.f, 0.f);
    }
    t.bias = *reinterpret_cast<const float2*>(wf + p.vf_fc1b + j);
    t.w2 = *reinterpret_cast<const float2*>(wf + p.vf_fc2w + j);
#pragma unroll
    for (int k = 0; k < 7; k++) t.wm[k] = *reinterpret_cast<const float2*>(wf + p.vf_fc1m + k * 128 + j);
    t.fc2b = wf[p.vf_fc2b];
}
__device__ __forceinline__ float value_tail_finish(const SpParams& p, const ValueTail& t) { return scvt::value_tail_compute(t, p.vf_ksplit); }

// cs_out / cs_valid: the control block as this function leaves it, handed to dev_select in registers (a reload would be
// a load of words stored a few instructions earlier); not valid after a ply transition
template <bool SC_ST = true>
__device__ __forceinline__ void dev_expand(SpParams& p, int g, int lane, Position* s_np_p, GameCtl& cs_out, bool& cs_valid) {
#pragma clang fp contract(off)
    Position& s_np = *s_np_p;
    GameCtl& c = p.ctl[g];
    // First round trip: EVERYTHING whose address depends on the game slot only -- the control block (one 64-byte
    // fetch), the recorded path, the leaf's priors and legal moves, the slot counters, the value partials.  Order
    // matters: the loads are issued with clamped indices and no guards, and the control block is moved to SGPRs only
    // AFTER the last of them -- a readfirstlane right behind its load (or a load under `cond ? load : 0`, which becomes
    // a branch around the load with its own wait) parks the wave for a full round trip before the next load is even
    // issued: the kernel used to start with three serialised trips (control block, path, the rest).
    // ... and the kernel arguments those addresses are made of are fetched TOGETHER: left to itself the compiler reads each of
    // them (scattered over the argument block's cache lines) where it is first used, behind a wait of its own -- 3 k cycles
    // passed between the entry of this function and the issue of its last load (tools/dbg_expand.py).  The empty asm statement
    // wants them all in SGPRs at one point: one batch of scalar loads, one wait.
    {
        const void *a0 = p.ctl, *a1 = p.path, *a2 = p.prior, *a3 = p.legal_mv, *a4 = p.slot_cnt, *a5 = p.vpart, *a6 = p.vf_w, *a7 = p.meta;
        const int i0 = p.max_depth, i1 = p.vf_fused, i2 = p.n_slots, i3 = p.vf_ksplit;
        const uint32_t u0 = p.vf_fc1b, u1 = p.vf_fc1m, u2 = p.vf_fc2w, u3 = p.vf_fc2b;
        asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(u0), "s"(u1),
                     "s"(u2), "s"(u3));
    }
    const GameCtl craw = c;
    const int pth_raw = p.path[(size_t)g * p.max_depth + (lane < p.max_depth ? lane : p.max_depth - 1)];
    float prv[4];
    uint16_t lmv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = lane + 64 * k, ic = i < MAXC ? i : MAXC - 1;
        prv[k] = p.prior[(size_t)g * MAXC + ic];
        lmv[k] = p.legal_mv[(size_t)g * MAXC + ic];
    }
    const unsigned long long sc_sims = p.slot_cnt[(size_t)g * 2], sc_evals = p.slot_cnt[(size_t)g * 2 + 1];
    ValueTail vt;
    if (p.vf_fused) value_tail_issue(p, g, lane, vt);   // used when the leaf turns out to be a network evaluation
    __builtin_amdgcn_sched_barrier(0);
    SC_XSTAMP(16);
    const GameCtl cs = uniform(craw);
    SC_XSTAMP(17);
    const int pth = lane < p.max_depth ? pth_raw : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = lane + 64 * k;
        prv[k] = i < MAXC ? prv[k] : 0.f;
        lmv[k] = i < MAXC ? lmv[k] : (uint16_t)0;
    }
    cs_out = cs;
    cs_valid = true;
    if (cs.status == ST_PENDING) {   // waiting for its trace-ring row (try_start_game)
        try_start_game(p, g, lane, cs.game_id - p.first_game_id);
        cs_valid = false;
        return;
    }
    if (cs.status != ST_ACTIVE || cs.leaf_kind == LK_NONE) return;
    const size_t nb = (size_t)g * p.node_cap;
    int32_t* N = p.N + nb;
    float* W = p.W + nb;
    float* P = p.P + nb;
    float* U = p.U + nb;
    uint16_t* MV = p.MV + nb;
    NodeHdr* H = p.H + nb;
    const int32_t* path = p.path + (size_t)g * p.max_depth;
    Position* hist = p.hist + (size_t)g * p.hist_cap;
    Position* tpos = p.tpos + (size_t)g * p.tpos_cap;

    const int leaf = cs.leaf, kind = cs.leaf_kind, plen = cs.path_len;
    float value = cs.leaf_value;
    int n_nodes = cs.n_nodes, n_exp = cs.n_exp;
    uint32_t err = 0;
    // Everything whose address depends on the game only (the recorded path, the leaf's priors and legal moves) was
    // requested together with the control block above; the statistics of the path nodes are requested now, before
    // the value tail, so the whole expansion costs two L2 round trips instead of five dependent ones.
    const bool inpath = lane < plen;
    int n0 = 0;
    float w0 = 0.f;
    if (inpath) {
        n0 = N[pth];
        w0 = W[pth];
    }
    if (kind == LK_EVAL) {
        value = p.vf_fused ? value_tail_finish(p, vt) : p.value[g];
        SC_XSTAMP(18);
        int n = cs.n_legal;
        if (n_nodes + n > p.node_cap || n_exp + 1 >= p.tpos_cap) {
            err = ERR_POOL_OVERFLOW;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int i = lane + 64 * k;
                if (i < n) {
                    int id = n_nodes + i;
                    N[id] = 0;
                    W[id] = 0.0f;
                    P[id] = prv[k];
                    U[id] = 0.0f;
                    MV[id] = lmv[k];
                    H[id] = NodeHdr{-1, 0, 0};
                }
            }
            if (lane == 0) H[leaf] = NodeHdr{n_nodes, (uint16_t)n, (uint16_t)n_exp};
            n_nodes += n;
            n_exp += 1;
        }
    } else if (kind == LK_TERM_NEW) {
        if (lane == 0) H[leaf] = NodeHdr{value == 0.0f ? -2 : value > 0.0f ? -3 : -4, 0, 0};
    }
    SC_XSTAMP(19);
    // backward (mcts.rs:90-98): every node of the path, root included
    if (inpath) {
        N[pth] = n0 + 1;
        W[pth] = w0 + value;
    }
    for (int d = lane + 64; d < plen; d += 64) {
        int nd = path[d];
        N[nd] += 1;
        W[nd] += value;
    }
    wave_sync();
    int sim = cs.sim + 1;
    if (lane == 0) {
        c.n_nodes = n_nodes;
        c.n_exp = n_exp;
        c.sim = sim;
        c.leaf_kind = LK_NONE;
        if (err) {
            c.err = cs.err | err;
            atomicOr(&p.cnt->err, (int)err);
        }
        // per-slot counters (summed by the host): 256 waves adding to ONE global counter at the same moment serialise
        // in the L2 atomic unit, and this wave's next loads queue behind its own atomics (in-order vmcnt)
        p.slot_cnt[(size_t)g * 2] = sc_sims + 1ULL;
        if (kind == LK_EVAL) p.slot_cnt[(size_t)g * 2 + 1] = sc_evals + 1ULL;
    }
    SC_XSTAMP(20);
    cs_out.n_nodes = n_nodes;
    cs_out.n_exp = n_exp;
    cs_out.sim = sim;
    cs_out.leaf_kind = LK_NONE;
    cs_out.err = cs.err | err;
    const int budget = p.rollout_factor > 0.f ? cs.rollout_cur : p.rollout;
    if (sim < budget) return;
    cs_valid = false;

    // ---------------- end of this ply's search (main.rs:198-233)
    wave_sync();
    const int ply = cs.ply;
    const NodeHdr h0 = H[0];
    const int nc = h0.nc, fc = h0.fc;
    if (nc == 0 || budget == 0) {
        // mcts::step -> None: no children => no legal moves (main.rs:213-216), or a --rollout-factor budget of 0
        // simulations (the reference then searches nothing and the root stays childless)
        HistChain hc{hist};
        int winner = -1;
        int term = outcome_claim_draw(hc, ply, &winner);
        finish_game(p, g, lane, term != T_NONE, term, winner);
        return;
    }
    const int ts = cs.trace_slot;
    const size_t tstep = (size_t)ts * p.num_steps + (size_t)(ply - cs.start_ply);
    for (int i = lane; i < nc; i += 64) {
        p.t_cmove[tstep * MAXC + i] = MV[fc + i];
        p.t_cn[tstep * MAXC + i] = N[fc + i];
        p.t_cq[tstep * MAXC + i] = W[fc + i];
        p.t_cu[tstep * MAXC + i] = U[fc + i];
    }
    // mcts::step (mcts.rs:298-317)
    float temperature = (ply - cs.start_ply) < p.temp_switch ? 1.0f : p.temperature;
    int choice = 0;
    if (temperature == 0.0f) {
        int bn = -1, bi = 0x7fffffff;
        for (int i = lane; i < nc; i += 64) {
            int n = N[fc + i];
            if (n > bn) {  // first maximum within the lane (indices increase)
                bn = n;
                bi = i;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            int on = __shfl_xor(bn, o, 64), oi = __shfl_xor(bi, o, 64);
            if (on > bn || (on == bn && oi < bi)) {
                bn = on;
                bi = oi;
            }
        }
        choice = bi;
        if (p.tie_random) {
            // NNPlayer::bestmove (play.rs:268-277): uniform among the maxima; k = floor(u * count), in index order
            int cnt = 0;
            for (int i = 0; i < nc; i++) cnt += N[fc + i] == bn;
            float u = (float)(sc_rng(p.seed, cs.game_id, (uint64_t)ply, 1, 0) >> 40) / 16777216.0f;
            int k = (int)(u * (float)cnt);
            if (k >= cnt) k = cnt - 1;
            for (int i = 0; i < nc; i++)
                if (N[fc + i] == bn && k-- == 0) {
                    choice = i;
                    break;
                }
        }
    } else {
        // WeightedIndex over N^(1/temp): sequential f32 cumulative sums, x = u*total,
        // index = number of cumulative weights (last excluded) <= x
        float power = 1.0f / temperature;
        float total = 0.0f;
        for (int i = 0; i < nc; i++) {
            float n = (float)N[fc + i];
            total += power == 1.0f ? n : powf(n, power);
        }
        float u = (float)(sc_rng(p.seed, cs.game_id, (uint64_t)ply, 1, 0) >> 40) / 16777216.0f;
        float x = u * total;
        float cum = 0.0f;
        int idx = 0;
        for (int i = 0; i < nc - 1; i++) {
            float n = (float)N[fc + i];
            cum += power == 1.0f ? n : powf(n, power);
            if (cum <= x) idx++;
        }
        choice = idx;
    }
    move_t mv = MV[fc + choice];
    if (lane == 0) {
        p.t_move[tstep] = mv;
        p.t_q[tstep] = W[0];
        p.t_nchild[tstep] = nc;
    }
    // advance the game line
    Position np = uniform(hist[ply]);
    make_move(np, (move_t)uniform((int)mv));
    if (lane == 0) s_np = np;
    wave_sync();
    {
        DevChain ch{hist, ply, tpos, nullptr, &s_np, ply + 1};
        uint8_t rf = rep_flags_wave(ch, ply + 1, np.key, lane);
        np.flags = (uint8_t)((np.flags & F_IRREV) | rf);
    }
    wave_sync();
    if (lane == 0) {
        hist[ply + 1] = np;
        atomicAdd(&p.cnt->plies_done, 1ULL);
    }
    wave_sync();
    const int new_ply = ply + 1;
    if (lane == 0) c.ply = new_ply;
    const int i_step = ply - cs.start_ply;  // the reference's loop index i
    if (i_step > p.outcome_gate) {         // main.rs:223-228
        HistChain hc{hist};
        int winner = -1;
        int term = outcome_claim_draw(hc, new_ply, &winner);
        if (term != T_NONE) {
            wave_sync();
            finish_game(p, g, lane, 1, term, winner);
            return;
        }
    }
    if (i_step + 1 >= p.num_steps || new_ply + 1 >= p.hist_cap) {  // loop ends: outcome stays null
        wave_sync();
        finish_game(p, g, lane, 0, 0, -1);
        return;
    }
    // fresh search tree rooted at the new position (chosen child is reset(), mcts.rs:319-323)
    if (lane == 0) {
        tpos[0] = np;
        N[0] = 0;
        W[0] = 0.0f;
        P[0] = 0.0f;
        U[0] = 0.0f;
        MV[0] = mv;
        H[0] = NodeHdr{-1, 0, 0};
        c.n_nodes = 1;
        c.n_exp = 1;
        c.sim = 0;
    }
}

// One launch per simulation step: finish the previous simulation of every game (value head tail, expand,
// backward, and at the end of a ply mcts::step + trace + outcome), then select the next leaf and encode it.
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_mcts(SpParams p, int do_expand, int do_select) {
    const int g = blockIdx.x, lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) int8_t s_stage[7168];
    __shared__ move_t s_moves[MAXC];
    __shared__ Position s_pos;
    __shared__ Position s_hist[8];
    __shared__ uint16_t s_ps[DEPTH_LDS];
    constexpr bool SC_ST = true;   // (k_mcts runs the flush, the timed samples and the synthetic evaluators: never the hot loop)
    SC_STAMP(0);
    GameCtl cs_pre{};
    bool cs_pre_valid = false;
    if (do_expand) {
        // No fence between the two halves: the block is ONE wavefront, whose vector-memory operations reach the cache
        // hierarchy in program order, so the selection below reads what the expansion above stored (statistics of the
        // path, control block, tree headers) without first waiting for every store to be acknowledged (~4 k cycles).
        dev_expand(p, g, lane, &s_pos, cs_pre, cs_pre_valid);
        __builtin_amdgcn_wave_barrier();
    }
    SC_STAMP(1);
    if (do_select) dev_select(p, g, lane, s_stage, s_moves, &s_pos, s_ps, s_hist, cs_pre, cs_pre_valid);
}
#endif

// ------------------------------------------------------------------ sc_selfplay_set_position
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_set_position(SpParams p, int g, const uint16_t* moves, int n_moves) {
    const int lane = threadIdx.x;
    GameCtl& c = p.ctl[g];
    Position* hist = p.hist + (size_t)g * p.hist_cap;
    Position* tpos = p.tpos + (size_t)g * p.tpos_cap;
    __shared__ Position s_np;
    Position cur;
    set_startpos(cur);
    cur.key = position_key(cur);
    if (lane == 0) hist[0] = cur;
    __syncthreads();
    for (int i = 0; i < n_moves && i + 1 < p.hist_cap; i++) {
        make_move(cur, moves[i]);
        if (lane == 0) s_np = cur;
        __syncthreads();
        DevChain ch{hist, i, tpos, nullptr, &s_np, i + 1};
        uint8_t rf = rep_flags_wave(ch, i + 1, cur.key, lane);
        cur.flags = (uint8_t)((cur.flags & F_IRREV) | rf);
        __syncthreads();
        if (lane == 0) hist[i + 1] = cur;
        __threadfence_block();
        __syncthreads();
    }
    if (lane == 0) {
        const size_t nb = (size_t)g * p.node_cap;
        tpos[0] = cur;
        p.N[nb] = 0;
        p.W[nb] = 0.0f;
        p.H[nb] = NodeHdr{-1, 0, 0};
        c.ply = n_moves;
        c.start_ply = n_moves;
        c.sim = 0;
        c.n_nodes = 1;
        c.n_exp = 1;
        c.leaf_kind = LK_NONE;
        c.rollout_cur = p.rollout;
        c.status = ST_ACTIVE;
    }
}
#endif

// ------------------------------------------------------------------ sc_encode_positions
// One wave per position: replay the move list from the start position (validating every move
// against the legal-move generator), then produce the NN input, the legal moves + action indices
// and outcome(claim_draw=True).  hist scratch: [n][hist_cap] Positions.
// move_len (optional): position g replays moves[move_off[g] .. move_off[g] + move_len[g]) -- prefixes of one game
// share their start (used by the training-tensor encoder: one position per ply).
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_encode_positions(int n_pos, const uint16_t* moves, const uint32_t* move_off,
                                                         const uint32_t* move_len, Position* hist_all, int hist_cap, int8_t* boards, int32_t* meta,
                                                         uint16_t* legal_mv, uint16_t* legal_idx, int32_t* n_legal,
                                                         int32_t* outcome) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= n_pos) return;
    __shared__ __attribute__((aligned(16))) int8_t s_stage[7168];
    __shared__ move_t s_moves[MAXC];
    __shared__ Position s_np;
    Position* hist = hist_all + (size_t)g * hist_cap;
    const uint16_t* mv = moves + move_off[g];
    int nm = move_len ? (int)move_len[g] : (int)(move_off[g + 1] - move_off[g]);
    Position cur;
    set_startpos(cur);
    cur.key = position_key(cur);
    if (lane == 0) hist[0] = cur;
    __syncthreads();
    int status = 0;
    int played = 0;
    for (int i = 0; i < nm && i + 1 < hist_cap; i++) {
        int nlm = 0;
        gen_legal_wave(cur, s_moves, lane, nlm);
        __syncthreads();
        bool found = false;
        for (int k = 0; k < nlm; k++)
            if (s_moves[k] == mv[i]) found = true;
        __syncthreads();
        if (!found) {
            status = -(i + 1);
            break;
        }
        make_move(cur, mv[i]);
        if (lane == 0) s_np = cur;
        __syncthreads();
        DevChain ch{hist, i, hist, nullptr, &s_np, i + 1};
        uint8_t rf = rep_flags_wave(ch, i + 1, cur.key, lane);
        cur.flags = (uint8_t)((cur.flags & F_IRREV) | rf);
        __syncthreads();
        if (lane == 0) hist[i + 1] = cur;
        __threadfence_block();
        __syncthreads();
        played = i + 1;
    }
    HistChain hc{hist};
    int n = 0;
    bool in_check = gen_legal_wave(cur, s_moves, lane, n);
    __syncthreads();
    if (legal_mv)
        for (int i = lane; i < n; i += 64) legal_mv[(size_t)g * MAXC + i] = s_moves[i];
    if (legal_idx)
        for (int i = lane; i < n; i += 64) legal_idx[(size_t)g * MAXC + i] = (uint16_t)move_index(s_moves[i], cur.turn);
    if (n_legal && lane == 0) n_legal[g] = n;
    if (boards) {
        int32_t mtmp_dummy[8];
        (void)mtmp_dummy;
        __shared__ int32_t s_meta[8];
        __shared__ Position s_hist[8];
        stage_history(hc, played, lane, s_hist);
        __syncthreads();
        encode_wave(s_hist, played < 7 ? played + 1 : 8, lane, s_stage, boards + (size_t)g * 7168, s_meta);
        __syncthreads();
        if (meta && lane < 7) meta[(size_t)g * 7 + lane] = s_meta[lane];
    } else if (meta && lane == 0) {
        int32_t m[7];
        encode_meta(cur, m);
        for (int k = 0; k < 7; k++) meta[(size_t)g * 7 + k] = m[k];
    }
    if (outcome) {
        int winner = -1;
        int term = outcome_claim_draw(hc, played, &winner);
        if (lane == 0) {
            outcome[(size_t)g * 4 + 0] = term;
            outcome[(size_t)g * 4 + 1] = winner;
            outcome[(size_t)g * 4 + 2] = in_check ? 1 : 0;
            outcome[(size_t)g * 4 + 3] = status;
        }
    }
}
#endif

// ------------------------------------------------------------------ trace replay for the training-tensor encoder
// sc_encode_steps needs every ply of every game.  Replaying each ply's prefix in its own wave (k_encode_positions) is
// O(plies^2) make_move + move generations per game; here a game is walked ONCE by one wave -- make_move, transposition key,
// repetition flags, one 80-byte record per ply -- and the plies are then encoded in parallel from those records
// (k_encode_plies).  The walk does not validate the moves (that needs a move generation per ply: the per-ply kernel has
// one anyway, and k_steps_dist checks the played move against it); it only refuses moves make_move could not execute
// safely -- no piece of the mover on the from-square, an own piece on the target, a promotion code outside {0, N, B, R, Q}
// or on a non-pawn -- and leaves the position unchanged for those (the ply is then reported as illegal by the per-ply check,
// and the game's later plies are unspecified, include/sc_engine.h).
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_replay_raw(int n_games, const uint16_t* moves, const uint32_t* move_off, Position* hist_all,
                                                   int hist_cap) {
    // the only sequential part: one wave per game, board updates only (make_move_board: ~10 % of what a full make_move + repetition
    // scan per ply cost when this kernel did everything -- 1.9 us per ply, 194 us for 100-ply games)
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= n_games) return;
    Position* hist = hist_all + (size_t)g * hist_cap;
    const uint16_t* mv = moves + move_off[g];
    const int nm = (int)(move_off[g + 1] - move_off[g]);
    Position cur;
    set_startpos(cur);
    cur.key = 0;
    cur.flags = 0;
    if (lane == 0) hist[0] = cur;
    int mv64 = 0;   // the next 64 moves of the game, one per lane: one load per 64 plies instead of a dependent load per ply
    for (int i = 0; i < nm && i + 1 < hist_cap; i++) {
        if ((i & 63) == 0) mv64 = (i + lane < nm) ? (int)mv[i + lane] : 0;
        const move_t m = (move_t)__builtin_amdgcn_readlane(mv64, i & 63);
        const int from = mv_from(m), to = mv_to(m), promo = mv_promo(m);
        const bool ours = (occ_c(cur, cur.turn) & bit(from)) != 0, own_target = (occ_c(cur, cur.turn) & bit(to)) != 0;
        const bool promo_ok = promo == 0 || (promo >= 2 && promo <= 5 && (cur.pcs[PAWN] & bit(from)) != 0);
        if (ours && !own_target && promo_ok && from != to) make_move_board(cur, m);
        if (lane == 0) hist[i + 1] = cur;
    }
}

// transposition key of a position, lane = square: the same value as position_key() (XOR of the per-(piece, square) keys and the
// state key), a wave XOR instead of a 32-iteration scalar loop
__device__ inline bb_t position_key_wave(const Position& p, int lane, bool ep_legal) {
    bb_t h = 0;
    if ((all_occ(p) >> lane) & 1) h = psq_key(piece_type_at(p, lane), (int)((p.occ[WHITE] >> lane) & 1), lane);
    unsigned lo = (unsigned)h, hi = (unsigned)(h >> 32);
    for (int o = 32; o > 0; o >>= 1) {
        lo ^= (unsigned)__shfl_xor((int)lo, o, 64);
        hi ^= (unsigned)__shfl_xor((int)hi, o, 64);
    }
    return (((bb_t)hi << 32) | lo) ^ state_key(p.turn, p.castling, ep_legal ? p.ep : -1);
}

// one wave per ply, after k_replay_raw: the record's key, and F_IRREV of the move that led to it (python-chess is_irreversible on
// the position before: zeroing, castling rights reduced, or a legal en-passant capture was available)
__global__ __launch_bounds__(64) void k_ply_keys(int n, Position* hist_all, const uint32_t* hoff, const uint32_t* plen, const uint16_t* ply_move) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= n) return;
    Position* hist = hist_all + hoff[q];
    const int i = (int)plen[q];
    const Position pos = uniform(hist[i]);
    const bool epl = has_legal_ep(pos);
    const bb_t key = position_key_wave(pos, lane, epl);
    uint8_t fl = 0;
    if (i > 0) {
        const Position prev = uniform(hist[i - 1]);
        const move_t m = (move_t)uniform((int)ply_move[q - 1]);   // the game's previous ply is the previous ply of the batch
        fl = (is_zeroing(prev, m) || reduces_castling(prev, m) || has_legal_ep(prev)) ? F_IRREV : 0;
    }
    if (lane == 0) {
        hist[i].key = key;
        hist[i].flags = fl;
    }
}
// ... and, with every key and F_IRREV in place, the repetition flags (planes 12 / 13): is_repetition(2) / is_repetition(3)
__global__ __launch_bounds__(64) void k_ply_rep(int n, Position* hist_all, const uint32_t* hoff, const uint32_t* plen) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= n) return;
    Position* hist = hist_all + hoff[q];
    const int i = (int)plen[q];
    const HistChain ch{hist};
    const bb_t key0 = uniform(hist[i].key);
    const uint8_t rf = (uint8_t)__builtin_amdgcn_readfirstlane((int)rep_flags_wave(ch, i, key0, lane));
    // (a byte store beside the F_IRREV bit other waves' scans read: that bit does not change here)
    if (lane == 0 && rf) hist[i].flags = (uint8_t)((hist[i].flags & F_IRREV) | rf);
}

// one wave per ply: the position BEFORE the ply's move from the game's records -- legal moves (python-chess order), action
// indices, planes, meta.  hoff[p]: record index of the game's start position, plen[p]: moves played before the ply.
__global__ __launch_bounds__(64) void k_encode_plies(int n, const Position* hist_all, const uint32_t* hoff, const uint32_t* plen, int8_t* boards,
                                                     int32_t* meta, uint16_t* legal_mv, uint16_t* legal_idx, int32_t* n_legal) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= n) return;
    __shared__ __attribute__((aligned(16))) int8_t s_stage[7168];
    __shared__ move_t s_moves[MAXC];
    __shared__ int32_t s_meta[8];
    __shared__ Position s_hist[8];
    const HistChain hc{hist_all + hoff[g]};
    const int played = (int)plen[g];
    stage_history(hc, played, lane, s_hist);
    __syncthreads();
    const Position cur = s_hist[0];
    int nl = 0;
    gen_legal_wave(cur, s_moves, lane, nl);
    __syncthreads();
    for (int i = lane; i < MAXC; i += 64) {   // whole rows: the entries past n_legal are zero (include/sc_engine.h)
        legal_mv[(size_t)g * MAXC + i] = i < nl ? s_moves[i] : (move_t)0;
        legal_idx[(size_t)g * MAXC + i] = i < nl ? (uint16_t)move_index(s_moves[i], cur.turn) : (uint16_t)0;
    }
    if (lane == 0) n_legal[g] = nl;
    encode_wave(s_hist, played < 7 ? played + 1 : 8, lane, s_stage, boards + (size_t)g * 7168, s_meta);
    __syncthreads();
    if (lane < 7) meta[(size_t)g * 7 + lane] = s_meta[lane];
}
#endif

// ------------------------------------------------------------------ training tensors
// Per ply of a recorded game: libsmartchess.chess_encode_steps (reference src/lib.rs:46-128) on top of
// k_encode_positions (planes / meta / legal moves of the position BEFORE the ply's move):
//   * checks the reference's two panics: the searched children must be exactly the legal moves, the played move
//     must be legal -> flags[g] bit 0 / bit 1;
//   * dist[index(move)] = count / (sum + 1e-5), index by the REAL mover (lib.rs:85-92, 105-113);
//   * apply_mirror: the planes and dist do not change (the stored boards are rotated once at push and once more at
//     view, lib.rs:80-98 + chess.rs:827-842 -- asserted on the oracle's literal restatement); meta becomes that of
//     Board::rotate(): [!turn, fullmove + (turn==White), K(opp), Q(opp), K(mover), Q(mover), halfmove].
// One wavefront per ply; HBM-bound writer (18.7 KB of dist per ply).
#ifndef SC_NO_KERNELS
__global__ __launch_bounds__(64) void k_steps_dist(int n, const uint16_t* legal_mv, const int32_t* n_legal, const uint16_t* next_mv,
                                                   const uint16_t* child_mv, const uint32_t* child_n, const uint32_t* child_off,
                                                   int apply_mirror, int32_t* meta, float* dist, int32_t* flags) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= n) return;
    __shared__ move_t s_lm[MAXC];
    __shared__ int s_hit[MAXC];
    const int nl = n_legal[g];
    const uint32_t c0 = child_off[g];
    const int nc = (int)(child_off[g + 1] - c0);
    const int turn = meta[(size_t)g * 7];
    float4* dz = reinterpret_cast<float4*>(dist + (size_t)g * 4672);
    for (int i = lane; i < 4672 / 4; i += 64) dz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < MAXC; i += 64) {
        s_lm[i] = i < nl ? legal_mv[(size_t)g * MAXC + i] : (move_t)0;
        s_hit[i] = 0;
    }
    __syncthreads();
    const move_t nx = next_mv[g];
    int has_next = 0, bad = 0;
    uint32_t sum = 0;
    for (int i = lane; i < nl; i += 64) has_next |= (s_lm[i] == nx) ? 1 : 0;
    for (int i = lane; i < nc; i += 64) {
        const move_t m = child_mv[c0 + i];
        int k = -1;
        for (int j = 0; j < nl; j++)
            if (s_lm[j] == m) k = j;
        if (k < 0) bad = 1;
        else s_hit[k] = 1;     // benign same-value race between duplicates
        sum += child_n[c0 + i];
    }
    __syncthreads();
    for (int i = lane; i < nl; i += 64) bad |= s_hit[i] ? 0 : 1;   // with nc == nl this also catches duplicate children
    bad |= (nc != nl) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);   // u32 wrap-around, as the reference's u32 sum in release mode
    const float den = (float)sum + 1e-5f;
    for (int i = lane; i < nc; i += 64) {
        const int idx = move_index(child_mv[c0 + i], turn);
        if (idx >= 0) dist[(size_t)g * 4672 + idx] = (float)child_n[c0 + i] / den;
    }
    const bool any_bad = __ballot(bad) != 0, any_next = __ballot(has_next) != 0;
    if (lane == 0) {
        flags[g] = (any_bad ? 1 : 0) | (any_next ? 0 : 2);
        if (apply_mirror) {
            int32_t* m = meta + (size_t)g * 7;
            const int32_t t = m[0], k0 = m[2], q0 = m[3], k1 = m[4], q1 = m[5];
            m[0] = 1 - t;
            m[1] = m[1] + (t == 1 ? 1 : 0);
            m[2] = k1;
            m[3] = q1;
            m[4] = k0;
            m[5] = q0;
        }
    }
}
#endif

}  // namespace sc
