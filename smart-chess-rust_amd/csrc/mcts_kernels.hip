// mcts_kernels.hip -- translation unit of the search kernels.  Build with -ffp-contract=off
// (exact f32 PUCT arithmetic, see mcts_kernels.hpp).
#include "mcts_kernels.hpp"

#include "launchers.hpp"

namespace scl {
void init_slots(const sc::SpParams& p, hipStream_t s) { hipLaunchKernelGGL(sc::k_init_slots, dim3(p.n_slots), dim3(64), 0, s, p); }
void mcts(const sc::SpParams& p, int do_expand, int do_select, hipStream_t s) {
    hipLaunchKernelGGL(sc::k_mcts, dim3(p.n_slots), dim3(64), 0, s, p, do_expand, do_select);
}
void synth_eval(const sc::SpParams& p, hipStream_t s) { hipLaunchKernelGGL(sc::k_synth_eval, dim3(p.n_slots), dim3(64), 0, s, p); }
void debug_find_max(const float* d_u, int n, int* d_out, hipStream_t s) { hipLaunchKernelGGL(sc::k_debug_find_max, dim3(1), dim3(64), 0, s, d_u, n, d_out); }
void set_position(const sc::SpParams& p, int slot, const uint16_t* d_moves, int n_moves, hipStream_t s) {
    hipLaunchKernelGGL(sc::k_set_position, dim3(1), dim3(64), 0, s, p, slot, d_moves, n_moves);
}
void encode_positions(int n_pos, const uint16_t* d_moves, const uint32_t* d_move_off, const uint32_t* d_move_len, sc::Position* d_hist,
                      int hist_cap, int8_t* boards, int32_t* meta, uint16_t* legal_mv, uint16_t* legal_idx, int32_t* n_legal,
                      int32_t* outcome, hipStream_t s) {
    hipLaunchKernelGGL(sc::k_encode_positions, dim3(n_pos), dim3(64), 0, s, n_pos, d_moves, d_move_off, d_move_len, d_hist, hist_cap,
                       boards, meta, legal_mv, legal_idx, n_legal, outcome);
}
void replay_games(int n_games, int n_plies, const uint16_t* d_moves, const uint32_t* d_move_off, sc::Position* d_hist, int hist_cap,
                  const uint32_t* d_hoff, const uint32_t* d_plen, hipStream_t s) {
    if (n_games <= 0 || n_plies <= 0) return;
    hipLaunchKernelGGL(sc::k_replay_raw, dim3(n_games), dim3(64), 0, s, n_games, d_moves, d_move_off, d_hist, hist_cap);
    hipLaunchKernelGGL(sc::k_ply_keys, dim3(n_plies), dim3(64), 0, s, n_plies, d_hist, d_hoff, d_plen, d_moves);
    hipLaunchKernelGGL(sc::k_ply_rep, dim3(n_plies), dim3(64), 0, s, n_plies, d_hist, d_hoff, d_plen);
}
void encode_plies(int n, const sc::Position* d_hist, const uint32_t* d_hoff, const uint32_t* d_plen, int8_t* boards, int32_t* meta,
                  uint16_t* legal_mv, uint16_t* legal_idx, int32_t* n_legal, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(sc::k_encode_plies, dim3(n), dim3(64), 0, s, n, d_hist, d_hoff, d_plen, boards, meta, legal_mv, legal_idx, n_legal);
}
void steps_dist(int n, const uint16_t* legal_mv, const int32_t* n_legal, const uint16_t* next_mv, const uint16_t* child_mv,
                const uint32_t* child_n, const uint32_t* child_off, int apply_mirror, int32_t* meta, float* dist, int32_t* flags,
                hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(sc::k_steps_dist, dim3(n), dim3(64), 0, s, n, legal_mv, n_legal, next_mv, child_mv, child_n, child_off,
                       apply_mirror, meta, dist, flags);
}
}  // namespace scl
