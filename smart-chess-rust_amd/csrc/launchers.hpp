// launchers.hpp -- host entry points of the two kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "mcts_types.hpp"
#include "nn_types.hpp"

namespace scl {
// mcts_kernels.hip (compiled with -ffp-contract=off)
void init_slots(const sc::SpParams& p, hipStream_t s);
void mcts(const sc::SpParams& p, int do_expand, int do_select, hipStream_t s);
void synth_eval(const sc::SpParams& p, hipStream_t s);
void debug_find_max(const float* d_u, int n, int* d_out, hipStream_t s);
void set_position(const sc::SpParams& p, int slot, const uint16_t* d_moves, int n_moves, hipStream_t s);
void encode_positions(int n_pos, const uint16_t* d_moves, const uint32_t* d_move_off, const uint32_t* d_move_len, sc::Position* d_hist,
                      int hist_cap, int8_t* boards, int32_t* meta, uint16_t* legal_mv, uint16_t* legal_idx, int32_t* n_legal,
                      int32_t* outcome, hipStream_t s);
// (all plies of all games: d_hoff / d_plen = record offset of the ply's game and moves played before the ply; d_moves[q] = ply q's move)
void replay_games(int n_games, int n_plies, const uint16_t* d_moves, const uint32_t* d_move_off, sc::Position* d_hist, int hist_cap,
                  const uint32_t* d_hoff, const uint32_t* d_plen, hipStream_t s);
void encode_plies(int n, const sc::Position* d_hist, const uint32_t* d_hoff, const uint32_t* d_plen, int8_t* boards, int32_t* meta,
                  uint16_t* legal_mv, uint16_t* legal_idx, int32_t* n_legal, hipStream_t s);
void steps_dist(int n, const uint16_t* legal_mv, const int32_t* n_legal, const uint16_t* next_mv, const uint16_t* child_mv,
                const uint32_t* child_n, const uint32_t* child_off, int apply_mirror, int32_t* meta, float* dist, int32_t* flags,
                hipStream_t s);
// nn_kernels.hip
const char* nn_init();  // sets kernel attributes; returns error text or nullptr
size_t tower_lds_bytes(int C);
bool tower_variant_available(int C, bool tower32);   // production builds carry one tower kernel per trunk width
void tower(const scnn::TowerArgs& a, hipStream_t s);
void value_fc1(const scnn::Fc1Args& a, hipStream_t s);
// step_kernels.hip: search wave + tower in one launch (slot g = position g; a.n_pos must equal p.n_slots)
const char* step_init();
int step_blocks_per_cu(const scnn::NetLayout& net);
void step(const scnn::TowerArgs& a, const sc::SpParams& p, int do_expand, hipStream_t s);
void value_finish(const scnn::VfinArgs& a, hipStream_t s);
}  // namespace scl
