// mcts_types.hpp -- device-resident state of the self-play engine (shared by host code and kernels).
#pragma once
#include <stdint.h>

#include "chess_rules.hpp"

namespace sc {

constexpr int MAXC = 224;  // SC_MAX_MOVES
enum { ST_IDLE = 0, ST_ACTIVE = 1, ST_FINISHED = 2, ST_PENDING = 3 };   // PENDING: a game id is drawn, its trace-ring row is still in use
enum { TR_FREE = 0, TR_LIVE = 1, TR_DONE = 2 };   // TraceHdr::state
enum { SYNTH_HASH = 1, SYNTH_COARSE = 2, SYNTH_UNIFORM = 3 };   // SpParams::evaluator of the synthetic evaluators (sc_engine.h)
enum { LK_NONE = 0, LK_EVAL = 1, LK_TERM_NEW = 2, LK_TERM_CACHED = 3 };
enum { ERR_NONFINITE_UCT = 1, ERR_POOL_OVERFLOW = 2, ERR_BAD_MOVE_INDEX = 4, ERR_DEPTH_OVERFLOW = 8, ERR_HELPER_TIMEOUT = 16, ERR_HANDOFF_TIMEOUT = 32 };

struct GameCtl {
    int32_t status, ply, sim, n_nodes, n_exp;
    int32_t leaf, path_len, leaf_kind, n_legal;
    float leaf_value;
    uint32_t err;
    int32_t trace_slot;
    uint64_t game_id;
    int32_t start_ply;  // ply index at which this slot's game started searching (sc_selfplay_set_position)
    int32_t rollout_cur;  // --rollout-factor: this ply's simulation budget min(300, n_legal * factor) (src/main.rs:175-176)
};

// per-node header, fetched in one 8-byte load (and, for all children of a node, in the same round trip as their
// statistics, so that a descent costs one dependent memory round trip per level)
struct NodeHdr {
    int32_t fc;   // >=0 first child; -1 unexpanded; -2/-3/-4 terminal with value 0/+1/-1
    uint16_t nc;  // number of children
    uint16_t ps;  // tpos slot of an expanded node
};

struct TraceHdr {
    int32_t n_steps, has_outcome, termination, winner;
    uint64_t game_id;
    int32_t state, pad;   // TR_FREE / TR_LIVE / TR_DONE
};

struct Counters {
    unsigned long long next_game, sims_done, nn_evals, plies_done;
    int32_t games_finished, err;
};

struct SpParams {
    int n_slots, rollout, num_steps, temp_switch, with_noise, outcome_gate, evaluator, external_noise;
    int tie_random;        // temperature 0: uniformly random child among the most visited (match play, src/play.rs:268-277)
    int trace_hold;        // trace ring: a finished trace is kept until the host has released it (sc_selfplay_poll); 0: overwritten
    float rollout_factor;  // > 0: per-ply budget min(300, n_legal * factor) instead of `rollout` (src/main.rs:175-176)
    uint64_t synth_salt;   // synthetic evaluator: second deterministic "player" (match tests)
    float cpuct, temperature, epsilon;
    uint64_t seed, first_game_id;
    int node_cap, max_depth, hist_cap, tpos_cap, trace_cap, total_games;
    GameCtl* ctl;
    Position* hist;   // [slot][hist_cap]
    Position* tpos;   // [slot][tpos_cap]
    int32_t* path;    // [slot][max_depth]
    int32_t* N;
    float* W;
    float* P;
    float* U;
    uint16_t* MV;
    NodeHdr* H;
    int8_t* boards;   // [slot][7168]
    int32_t* meta;    // [slot][8]
    uint16_t* legal_mv;   // [slot][MAXC]
    uint16_t* legal_idx;  // [slot][MAXC]
    int32_t* n_legal;     // [slot]
    float* prior;         // [slot][MAXC]
    float* value;         // [slot]
    float* noise;         // [slot][MAXC]
    TraceHdr* thdr;       // [trace_cap]
    uint16_t* t_move;     // [trace_cap][num_steps]
    float* t_q;
    int32_t* t_nchild;
    uint16_t* t_cmove;    // [trace_cap][num_steps][MAXC]
    int32_t* t_cn;
    float* t_cq;
    float* t_cu;
    Counters* cnt;
    unsigned long long* slot_cnt;   // [n_slots][2] simulations / network evaluations per slot (no contended atomics on the hot path)
    // fused value-head tail (NET evaluator): split-K partials of value_head.ffn.0 + fp32 parameters
    int vf_fused, vf_ksplit;
    const float* vpart;
    const float* vf_w;
    uint32_t vf_fc1b, vf_fc1m, vf_fc2w, vf_fc2b;
    unsigned long long* dbg_cycles;  // optional [slot][32] stamps of the last launch (developer aid): 0..7 the search's cycle stamps, 8.. experiment builds
};


}  // namespace sc
