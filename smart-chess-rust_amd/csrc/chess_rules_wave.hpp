// chess_rules_wave.hpp -- legal move generation with the 64 squares on the 64 lanes of a wavefront.
//
// gen_legal (chess_rules.hpp) is wave-uniform code: every lane walks the same pieces and emits the same ~30 moves one
// after the other on the scalar unit (~14 k cycles per leaf, the largest single item of the search kernel).  Here
// lane s owns square s: attack sets, pin masks and target sets are computed for all pieces at once, the number of
// moves per lane is turned into output positions by ONE packed suffix scan (lanes in descending order = python-chess's
// from-square order), and every lane writes its own moves.  The result -- the list AND its order -- is identical to
// gen_legal (python-chess generate_legal_moves: non-pawn pieces by from-square descending / to-square descending,
// castling, pawn captures, single pushes, double pushes, en passant; evasions: king steps first); the parity tests
// compare both against the oracle on ~10^5 positions.
#pragma once
#include <hip/hip_runtime.h>

#include "chess_rules.hpp"

namespace sc {

// Cross-lane steps are DPP moves inside each row of 16 lanes plus readlanes between the four rows -- plain VALU /
// scalar work; the xor / shift shuffles they replace were 6-12 dependent trips through the LDS crossbar per call.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);   // lanes without a source read 0
}
__device__ __forceinline__ unsigned wave_or32(unsigned v) {
    v |= dpp_u<0xB1>(v);    // quad_perm [1,0,3,2]
    v |= dpp_u<0x4E>(v);    // quad_perm [2,3,0,1]
    v |= dpp_u<0x141>(v);   // row_half_mirror
    v |= dpp_u<0x140>(v);   // row_mirror: every lane holds its row's OR
    return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) | __builtin_amdgcn_readlane((int)v, 32) |
                      __builtin_amdgcn_readlane((int)v, 48));
}
// OR over the wave; the result is wave-uniform (SGPRs)
__device__ __forceinline__ bb_t wave_or64(bb_t v) { return ((bb_t)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v); }
__device__ __forceinline__ bb_t uniform64(bb_t v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((bb_t)hi << 32) | lo;
}
// sum of x over the lanes ABOVE this one (lane 63 gets 0), and the wave total in `total`: inclusive suffix scan inside
// each row (row_shl:n -- lane i reads lane i+n of its row, 0 past the row's end), then the totals of the rows above
__device__ __forceinline__ unsigned wave_suffix_excl(unsigned x, int lane, unsigned& total) {
    unsigned s = x;
    s += dpp_u<0x101>(s);   // row_shl:1
    s += dpp_u<0x102>(s);   // row_shl:2
    s += dpp_u<0x104>(s);   // row_shl:4
    s += dpp_u<0x108>(s);   // row_shl:8
    const unsigned t0 = (unsigned)__builtin_amdgcn_readlane((int)s, 0), t1 = (unsigned)__builtin_amdgcn_readlane((int)s, 16);
    const unsigned t2 = (unsigned)__builtin_amdgcn_readlane((int)s, 32), t3 = (unsigned)__builtin_amdgcn_readlane((int)s, 48);
    const int row = lane >> 4;
    s += (row < 1 ? t1 : 0u) + (row < 2 ? t2 : 0u) + (row < 3 ? t3 : 0u);
    total = t0 + t1 + t2 + t3;
    return s - x;
}

struct WaveGen {
    int us, king;
    bb_t ours, theirs, occ, blockers, danger;
};

// attack set of the piece (type t, colour c) on this lane's square: ONE rook-line and ONE bishop-line evaluation per
// wave (a switch over the piece type would run every case in turn under exec masks, the queen's twice over)
__device__ __forceinline__ bb_t piece_attacks_lane(int sq, int t, int c, bb_t occ) {
    const bb_t b = bit(sq);
    bb_t a = 0;
    if (t == PAWN) a = pawn_attacks_bb(c, b);
    if (t == KNIGHT) a = knight_attacks_bb(b);
    if (t == KING) a = king_attacks_bb(b);
    if (t == ROOK || t == QUEEN) a |= rook_attacks(sq, occ);
    if (t == BISHOP || t == QUEEN) a |= bishop_attacks(sq, occ);
    return a;
}

// one category-complete pass: moves whose from-square is in from_mask and to-square in to_mask, appended at s_moves[n]
__device__ inline int gen_pseudo_wave(const Position& p, const WaveGen& g, int t, bb_t att, bb_t from_mask, bb_t to_mask, move_t* s_moves,
                                      int n, int lane) {
    const int us = g.us;
    const bb_t me = bit(lane);
    const bool mine = (g.ours & me & from_mask) != 0;   // t / att: type and attack set of the piece on this square
    const bool pinned = (g.blockers & me) != 0;
    bb_t pinline = BB_ALL;
    if (pinned && g.king >= 0) pinline = line_through(lane, g.king);
    // ---- non-pawn pieces: targets of the piece on this lane's square
    bb_t moves = 0;
    if (mine && t != PAWN) {
        moves = att & ~g.ours & to_mask;
        moves &= (lane == g.king) ? ~g.danger : pinline;
    }
    // ---- pawn captures from this square
    bb_t caps = 0;
    const bool mypawn = mine && t == PAWN;
    if (mypawn) caps = att & g.theirs & to_mask & pinline;
    // ---- pawn pushes TO this square (python-chess walks the target squares in descending order)
    const int f1 = lane + (us ? -8 : 8), f2 = lane + (us ? -16 : 16);
    bool single = false, dbl = false;
    if (!(g.occ & me) && (to_mask & me)) {
        if (f1 >= 0 && f1 < 64) {
            const bb_t b1 = bit(f1);
            if (p.pcs[PAWN] & g.ours & from_mask & b1) {
                single = !((g.blockers & b1) && g.king >= 0 && !(line_through(f1, g.king) & me));
            } else if (!(g.occ & b1) && f2 >= 0 && f2 < 64 && (lane >> 3) == (us ? 3 : 4)) {
                const bb_t b2 = bit(f2);
                if (p.pcs[PAWN] & g.ours & from_mask & b2) dbl = !((g.blockers & b2) && g.king >= 0 && !(line_through(f2, g.king) & me));
            }
        }
    }
    const bool promo_to = (lane >> 3) == 0 || (lane >> 3) == 7;
    const bool promo_from = (lane >> 3) == (us ? 6 : 1);   // a pawn capturing from here lands on the last rank
    const unsigned c1 = (unsigned)popcnt(moves);
    const unsigned c3 = (unsigned)popcnt(caps) * (promo_from ? 4u : 1u);
    const unsigned c4 = single ? (promo_to ? 4u : 1u) : 0u;
    const unsigned c5 = dbl ? 1u : 0u;
    // castling sits between the piece moves and the pawn captures (uniform, at most two moves)
    unsigned tot = 0;
    const unsigned off = wave_suffix_excl(c1 | (c3 << 8) | (c4 << 16) | (c5 << 24), lane, tot);   // every category total <= 218 < 256
    const int n1 = (int)(tot & 255), n3 = (int)((tot >> 8) & 255), n4 = (int)((tot >> 16) & 255), n5 = (int)(tot >> 24);
    int w = n + (int)(off & 255);
    while (moves) {
        const int to = msb(moves);
        moves ^= bit(to);
        if (w < MAX_MOVES) s_moves[w] = mk_move(lane, to, 0);
        w++;
    }
    int nn = n + n1;
    if (from_mask & p.pcs[KING]) {
        MoveList ml{s_moves, nn};
        GenCtx gc;
        gc.p = &p;
        gc.king = g.king;
        gc.blockers = g.blockers;
        gc.danger = g.danger;
        gc.out = &ml;
        gen_castling(gc, from_mask, to_mask);   // uniform: every lane writes the same one or two moves
        nn = ml.n;
    }
    w = nn + (int)((off >> 8) & 255);
    while (caps) {
        const int to = msb(caps);
        caps ^= bit(to);
        if (promo_from) {
            if (w + 3 < MAX_MOVES) {
                s_moves[w] = mk_move(lane, to, 5);
                s_moves[w + 1] = mk_move(lane, to, 4);
                s_moves[w + 2] = mk_move(lane, to, 3);
                s_moves[w + 3] = mk_move(lane, to, 2);
            }
            w += 4;
        } else {
            if (w < MAX_MOVES) s_moves[w] = mk_move(lane, to, 0);
            w++;
        }
    }
    nn += n3;
    if (single) {
        w = nn + (int)((off >> 16) & 255);
        if (promo_to) {
            if (w + 3 < MAX_MOVES) {
                s_moves[w] = mk_move(f1, lane, 5);
                s_moves[w + 1] = mk_move(f1, lane, 4);
                s_moves[w + 2] = mk_move(f1, lane, 3);
                s_moves[w + 3] = mk_move(f1, lane, 2);
            }
        } else if (w < MAX_MOVES) {
            s_moves[w] = mk_move(f1, lane, 0);
        }
    }
    nn += n4;
    if (dbl) {
        w = nn + (int)(off >> 24);
        if (w < MAX_MOVES) s_moves[w] = mk_move(f2, lane, 0);
    }
    nn += n5;
    if (p.ep >= 0) {
        MoveList ml{s_moves, nn};
        GenCtx gc;
        gc.p = &p;
        gc.king = g.king;
        gc.blockers = g.blockers;
        gc.danger = g.danger;
        gc.out = &ml;
        gen_ep(gc, from_mask, to_mask);
        nn = ml.n;
    }
    return nn < MAX_MOVES ? nn : MAX_MOVES;
}

// python-chess generate_legal_moves() into s_moves (LDS, capacity MAX_MOVES); p must be wave-uniform.  Returns the
// move count in n_out and whether the side to move is in check.  The caller synchronises before reading s_moves.
__device__ inline bool gen_legal_wave(const Position& p, move_t* s_moves, int lane, int& n_out) {
    WaveGen g;
    g.us = p.turn;
    g.ours = occ_c(p, g.us);
    g.theirs = occ_c(p, !g.us);
    g.occ = g.ours | g.theirs;
    const bb_t kbb = p.pcs[KING] & g.ours;
    // One attack-set evaluation per lane serves both sides: the opponent's pieces see the board with our king lifted
    // off (python-chess _attacked_for_king: sliders x-ray through it), our own pieces see the real occupancy.
    const bb_t me = bit(lane);
    const int t = piece_type_at(p, lane);
    const bool enemy = (g.theirs & me) != 0;
    bb_t att = 0;
    if (g.occ & me) att = piece_attacks_lane(lane, t, enemy ? !g.us : g.us, enemy ? (g.occ ^ kbb) : g.occ);
    if (!kbb) {
        g.king = -1;
        g.blockers = 0;
        g.danger = 0;
        n_out = gen_pseudo_wave(p, g, t, att, BB_ALL, BB_ALL, s_moves, 0, lane);
        return false;
    }
    g.king = msb(kbb);
    g.blockers = slider_blockers(p, g.king);
    // a slider's x-ray through the king adds squares BEHIND the king only, so "attacks the king" is unchanged
    const bool checks = enemy && (att & kbb) != 0;
    g.danger = uniform64(wave_or64(enemy ? att : 0));
    const bb_t checkers = (bb_t)__ballot(checks);
    if (!checkers) {
        n_out = gen_pseudo_wave(p, g, t, att, BB_ALL, BB_ALL, s_moves, 0, lane);
        return false;
    }
    // _generate_evasions: king steps first (to-square descending), then captures / interpositions for a single checker
    const bb_t kt = king_attacks_bb(kbb) & ~g.ours & ~g.danger;
    unsigned tot = 0;
    const unsigned off = wave_suffix_excl((kt & me) ? 1u : 0u, lane, tot);
    if ((kt & me) && off < (unsigned)MAX_MOVES) s_moves[off] = mk_move(g.king, lane, 0);
    int n = (int)tot;
    const int checker = msb(checkers);
    if (bit(checker) == checkers) {
        const bb_t target = between(g.king, checker) | checkers;
        n = gen_pseudo_wave(p, g, t, att, ~p.pcs[KING], target, s_moves, n, lane);
        if (p.ep >= 0 && !(bit(p.ep) & target)) {
            const int last_double = p.ep + (g.us ? -8 : 8);
            if (last_double == checker) {
                MoveList ml{s_moves, n};
                GenCtx gc;
                gc.p = &p;
                gc.king = g.king;
                gc.blockers = g.blockers;
                gc.danger = g.danger;
                gc.out = &ml;
                gen_ep(gc, BB_ALL, BB_ALL);
                n = ml.n;
            }
        }
    }
    n_out = n;
    return true;
}

}  // namespace sc
