// chess_rules.hpp -- bitboard chess rules for the MI355X self-play engine.
//
// Replaces the reference's rules oracle: python-chess 1.11.1 reached through pyo3 from
// src/chess.rs:665-803 (BoardState::{new,legal_moves,next,dup,outcome,turn,to_board}) and
// src/chess.rs:356-412 (Board extraction: is_repetition(2|3), castling rights, clocks).
// Everything is table-free arithmetic (o^(o-2r) line attacks with bit reversal, shift-based
// leaper attacks) so that a 64-lane wavefront can run it wave-uniformly without LDS or constant
// tables; the same source compiles for the host (g++) for CPU unit tests (perft etc.).
//
// Move order is python-chess's generate_legal_moves() order, because child order is visible in
// the trace file and in the reference's tie-breaks (src/mcts.rs:78-88, :309-311).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SC_HD __host__ __device__ inline
#else
#define SC_HD inline
#endif

namespace sc {

typedef uint64_t bb_t;
enum { PAWN = 0, KNIGHT = 1, BISHOP = 2, ROOK = 3, QUEEN = 4, KING = 5 };
enum { BLACK = 0, WHITE = 1 };

constexpr bb_t FILE_A = 0x0101010101010101ULL;
constexpr bb_t FILE_H = 0x8080808080808080ULL;
constexpr bb_t RANK_1 = 0xFFULL;
constexpr bb_t RANK_8 = 0xFF00000000000000ULL;
constexpr bb_t BB_ALL = ~0ULL;
constexpr int MAX_MOVES = 224;  // 218 is the known maximum; padded

// flags byte of Position
constexpr uint8_t F_REP2 = 1;   // is_repetition(2) at this position (plane 12, src/chess.rs:646)
constexpr uint8_t F_REP3 = 2;   // is_repetition(3)                    (plane 13, src/chess.rs:647)
constexpr uint8_t F_IRREV = 4;  // the move that led here was irreversible (python-chess is_irreversible)

struct Position {
    bb_t pcs[6];   // by piece type, both colours
    bb_t occ[2];   // by colour (occ[WHITE], occ[BLACK])
    bb_t key;      // hash of python-chess _transposition_key() (ep only if a legal ep capture exists)
    uint8_t turn;  // WHITE = 1
    uint8_t castling;  // bit0 h1, bit1 a1, bit2 h8, bit3 a8: rook squares that keep rights (clean)
    int8_t ep;         // python-chess ep_square: set after ANY double push, else -1
    uint8_t flags;
    uint16_t halfmove;
    uint16_t fullmove;
};
static_assert(sizeof(Position) == 80, "Position layout");

// move: from | to<<6 | promo<<12, promo in python-chess piece types (0 none, 2 N, 3 B, 4 R, 5 Q)
typedef uint16_t move_t;
SC_HD move_t mk_move(int from, int to, int promo) { return (move_t)(from | (to << 6) | (promo << 12)); }
SC_HD int mv_from(move_t m) { return m & 63; }
SC_HD int mv_to(move_t m) { return (m >> 6) & 63; }
SC_HD int mv_promo(move_t m) { return (m >> 12) & 7; }

// ------------------------------------------------------------------ bit helpers
SC_HD bb_t bit(int sq) { return 1ULL << sq; }
SC_HD int msb(bb_t b) { return 63 - __builtin_clzll(b); }
SC_HD int lsb(bb_t b) { return __builtin_ctzll(b); }
SC_HD int popcnt(bb_t b) { return __builtin_popcountll(b); }
SC_HD bb_t brev(bb_t x) {
#if defined(__clang__)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
#endif
}
SC_HD uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// ------------------------------------------------------------------ attack sets
SC_HD bb_t rank_mask(int sq) { return RANK_1 << (sq & 56); }
SC_HD bb_t file_mask(int sq) { return FILE_A << (sq & 7); }
SC_HD bb_t diag_mask(int sq) {
    const bb_t maindia = 0x8040201008040201ULL;
    int diag = 8 * (sq & 7) - (sq & 56);
    int nort = -diag & (diag >> 31);
    int sout = diag & (-diag >> 31);
    return (maindia >> sout) << nort;
}
SC_HD bb_t anti_mask(int sq) {
    const bb_t maindia = 0x0102040810204080ULL;
    int diag = 56 - 8 * (sq & 7) - (sq & 56);
    int nort = -diag & (diag >> 31);
    int sout = diag & (-diag >> 31);
    return (maindia >> sout) << nort;
}
// attacks along one line (mask includes sq) for a slider on sq: both rays up to and including the first blocker
SC_HD bb_t line_attacks(bb_t occ, int sq, bb_t mask) {
    bb_t s = bit(sq);
    bb_t m = mask ^ s;
    bb_t o = occ & m;
    bb_t f = o - s;
    bb_t r = brev(o) - bit(63 - sq);
    return (f ^ brev(r)) & m;
}
SC_HD bb_t rook_attacks(int sq, bb_t occ) { return line_attacks(occ, sq, rank_mask(sq)) | line_attacks(occ, sq, file_mask(sq)); }
SC_HD bb_t bishop_attacks(int sq, bb_t occ) { return line_attacks(occ, sq, diag_mask(sq)) | line_attacks(occ, sq, anti_mask(sq)); }
SC_HD bb_t knight_attacks_bb(bb_t b) {
    bb_t l1 = (b >> 1) & 0x7f7f7f7f7f7f7f7fULL, l2 = (b >> 2) & 0x3f3f3f3f3f3f3f3fULL;
    bb_t r1 = (b << 1) & 0xfefefefefefefefeULL, r2 = (b << 2) & 0xfcfcfcfcfcfcfcfcULL;
    bb_t h1 = l1 | r1, h2 = l2 | r2;
    return (h1 << 16) | (h1 >> 16) | (h2 << 8) | (h2 >> 8);
}
SC_HD bb_t king_attacks_bb(bb_t b) {
    bb_t a = ((b << 1) & ~FILE_A) | ((b >> 1) & ~FILE_H);
    bb_t c = b | a;
    return a | (c << 8) | (c >> 8);
}
// squares attacked by pawns of `color` standing on b
SC_HD bb_t pawn_attacks_bb(int color, bb_t b) {
    return color ? (((b << 7) & ~FILE_H) | ((b << 9) & ~FILE_A)) : (((b >> 7) & ~FILE_A) | ((b >> 9) & ~FILE_H));
}
// the full line through a and b (edge to edge) or 0 -- python-chess ray()
SC_HD bb_t line_through(int a, int b) {
    bb_t bb = bit(b);
    if (rank_mask(a) & bb) return rank_mask(a);
    if (file_mask(a) & bb) return file_mask(a);
    if (diag_mask(a) & bb) return diag_mask(a);
    if (anti_mask(a) & bb) return anti_mask(a);
    return 0;
}
// squares strictly between a and b on a shared line, else 0 -- python-chess between()
SC_HD bb_t between(int a, int b) {
    if (a == b) return 0;
    int lo = a < b ? a : b, hi = a < b ? b : a;
    bb_t span = (bit(hi) - 1) & ~((bit(lo) << 1) - 1);
    return span & line_through(a, b);
}

// NOTE: Position members are only ever indexed with compile-time constants (or through the helpers below):
// a runtime index would pin the struct in scratch memory on the GPU instead of registers.
SC_HD int piece_type_at(const Position& p, int sq) {
    bb_t b = bit(sq);
    int r = -1;
#pragma unroll
    for (int t = 5; t >= 0; t--)
        if (p.pcs[t] & b) r = t;
    return r;
}
SC_HD bb_t occ_c(const Position& p, int color) { return color ? p.occ[1] : p.occ[0]; }
SC_HD void xor_occ(Position& p, int color, bb_t b) {
    p.occ[0] ^= color ? 0 : b;
    p.occ[1] ^= color ? b : 0;
}
SC_HD void xor_pcs(Position& p, int type, bb_t b) {
#pragma unroll
    for (int t = 0; t < 6; t++) p.pcs[t] ^= (t == type) ? b : 0;
}
SC_HD bb_t all_occ(const Position& p) { return p.occ[0] | p.occ[1]; }

// python-chess attackers_mask(color, square) under an explicit occupancy
SC_HD bb_t attackers_mask(const Position& p, int color, int sq, bb_t occ) {
    bb_t rq = p.pcs[ROOK] | p.pcs[QUEEN], bq = p.pcs[BISHOP] | p.pcs[QUEEN];
    bb_t a = (rook_attacks(sq, occ) & rq) | (bishop_attacks(sq, occ) & bq) | (knight_attacks_bb(bit(sq)) & p.pcs[KNIGHT]) |
             (king_attacks_bb(bit(sq)) & p.pcs[KING]) | (pawn_attacks_bb(!color, bit(sq)) & p.pcs[PAWN]);
    return a & occ_c(p, color) & occ;
}
SC_HD bb_t piece_attacks(const Position& /*p*/, int sq, int type, int color, bb_t occ) {
    switch (type) {
        case PAWN: return pawn_attacks_bb(color, bit(sq));
        case KNIGHT: return knight_attacks_bb(bit(sq));
        case BISHOP: return bishop_attacks(sq, occ);
        case ROOK: return rook_attacks(sq, occ);
        case QUEEN: return rook_attacks(sq, occ) | bishop_attacks(sq, occ);
        default: return king_attacks_bb(bit(sq));
    }
}

// ------------------------------------------------------------------ start position
SC_HD void set_startpos(Position& p) {
    p.pcs[PAWN] = 0x00FF00000000FF00ULL;
    p.pcs[KNIGHT] = 0x4200000000000042ULL;
    p.pcs[BISHOP] = 0x2400000000000024ULL;
    p.pcs[ROOK] = 0x8100000000000081ULL;
    p.pcs[QUEEN] = 0x0800000000000008ULL;
    p.pcs[KING] = 0x1000000000000010ULL;
    p.occ[WHITE] = 0x000000000000FFFFULL;
    p.occ[BLACK] = 0xFFFF000000000000ULL;
    p.turn = WHITE;
    p.castling = 15;
    p.ep = -1;
    p.flags = 0;
    p.halfmove = 0;
    p.fullmove = 1;
    p.key = 0;
}

// ------------------------------------------------------------------ en passant legality
SC_HD bb_t ep_capturers(const Position& p, bb_t from_mask) {
    if (p.ep < 0) return 0;
    int us = p.turn;
    bb_t rank = us ? (RANK_1 << 32) : (RANK_1 << 24);
    return p.pcs[PAWN] & occ_c(p, us) & from_mask & pawn_attacks_bb(!us, bit(p.ep)) & rank;
}
// exact: play the capture on the bitboards and look for attackers of our king
SC_HD bool ep_is_legal(const Position& p, int from) {
    int us = p.turn;
    bb_t kbb = p.pcs[KING] & occ_c(p, us);
    if (!kbb) return true;
    int king = msb(kbb);
    int capsq = p.ep + (us ? -8 : 8);
    bb_t occ = (all_occ(p) ^ bit(from) ^ bit(capsq)) | bit(p.ep);
    bb_t theirs = occ_c(p, !us) & ~bit(capsq);
    bb_t rq = (p.pcs[ROOK] | p.pcs[QUEEN]) & theirs, bq = (p.pcs[BISHOP] | p.pcs[QUEEN]) & theirs;
    bb_t a = (rook_attacks(king, occ) & rq) | (bishop_attacks(king, occ) & bq) | (knight_attacks_bb(kbb) & p.pcs[KNIGHT] & theirs) |
             (pawn_attacks_bb(us, kbb) & p.pcs[PAWN] & theirs) | (king_attacks_bb(kbb) & p.pcs[KING] & theirs);
    return a == 0;
}
SC_HD bool has_legal_ep(const Position& p) {
    bb_t c = ep_capturers(p, BB_ALL);
    if (p.ep >= 0 && (all_occ(p) & bit(p.ep))) return false;
    while (c) {
        int from = msb(c);
        c ^= bit(from);
        if (ep_is_legal(p, from)) return true;
    }
    return false;
}

// hash of _transposition_key(): pieces, turn, clean castling rights, ep square iff a legal ep exists.
// Zobrist-style: XOR of per-(piece,square) keys and one state key, so make_move can update it incrementally.
SC_HD bb_t psq_key(int type, int is_white, int sq) { return mix64(((uint64_t)(type + (is_white ? 0 : 6)) << 8) | (uint64_t)sq); }
SC_HD bb_t state_key(int turn, int castling, int ep_if_legal) {
    return mix64(0x10000ULL | (uint64_t)turn | ((uint64_t)castling << 1) | ((uint64_t)(ep_if_legal + 1) << 5));
}
SC_HD bb_t position_key(const Position& p) {
    bb_t h = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) {
        bb_t b = p.pcs[t];
        while (b) {
            int sq = lsb(b);
            b &= b - 1;
            h ^= psq_key(t, (int)((p.occ[WHITE] >> sq) & 1), sq);
        }
    }
    return h ^ state_key(p.turn, p.castling, has_legal_ep(p) ? p.ep : -1);
}

// ------------------------------------------------------------------ Board.push
SC_HD bool is_zeroing(const Position& p, move_t m) {
    bb_t touched = bit(mv_from(m)) ^ bit(mv_to(m));
    return (touched & p.pcs[PAWN]) || (touched & occ_c(p, !p.turn));
}
SC_HD uint8_t castle_bit_for_sq(int sq) { return sq == 7 ? 1 : sq == 0 ? 2 : sq == 63 ? 4 : sq == 56 ? 8 : 0; }
SC_HD bool reduces_castling(const Position& p, move_t m) {
    int f = mv_from(m), t = mv_to(m);
    bb_t touched = bit(f) ^ bit(t);
    if (p.castling & (castle_bit_for_sq(f) | castle_bit_for_sq(t))) return true;
    if ((p.castling & 3) && (touched & p.pcs[KING] & p.occ[WHITE])) return true;
    if ((p.castling & 12) && (touched & p.pcs[KING] & p.occ[BLACK])) return true;
    return false;
}
// python-chess is_irreversible(move), evaluated on the position BEFORE the move
SC_HD bool is_irreversible(const Position& p, move_t m) { return is_zeroing(p, m) || reduces_castling(p, m) || has_legal_ep(p); }

// Plays m on p (flags' REP bits are cleared; IRREV and key are set). Pure function of (p, m); p.key must be
// valid on entry (it is updated incrementally: <= 7 key terms instead of a 32-piece recompute).
SC_HD void make_move(Position& p, move_t m) {
    int from = mv_from(m), to = mv_to(m), promo = mv_promo(m);
    int us = p.turn, them = !us;
    bool old_ep_legal = has_legal_ep(p);
    bool zero = is_zeroing(p, m);
    bool irrev = zero || reduces_castling(p, m) || old_ep_legal;  // python-chess is_irreversible
    bb_t key = p.key ^ state_key(us, p.castling, old_ep_legal ? p.ep : -1);
    int old_ep = p.ep;
    p.ep = -1;
    p.halfmove = zero ? 0 : (uint16_t)(p.halfmove + 1);
    if (us == BLACK) p.fullmove++;
    bb_t fb = bit(from), tb = bit(to);
    int pt = piece_type_at(p, from);
    int cap = (occ_c(p, them) & tb) ? piece_type_at(p, to) : -1;
    // remove mover
    xor_pcs(p, pt, fb);
    xor_occ(p, us, fb);
    key ^= psq_key(pt, us, from);
    p.castling &= (uint8_t)~(castle_bit_for_sq(from) | castle_bit_for_sq(to));
    if (pt == KING) p.castling &= us ? (uint8_t)~3 : (uint8_t)~12;
    if (cap >= 0) {
        xor_pcs(p, cap, tb);
        xor_occ(p, them, tb);
        key ^= psq_key(cap, them, to);
    }
    if (pt == PAWN) {
        int diff = to - from;
        if (diff == 16 && (from >> 3) == 1) p.ep = (int8_t)(from + 8);
        else if (diff == -16 && (from >> 3) == 6) p.ep = (int8_t)(from - 8);
        else if (to == old_ep && (diff == 7 || diff == 9 || diff == -7 || diff == -9) && cap < 0) {
            int csq = old_ep + (us ? -8 : 8);
            bb_t cb = bit(csq);
            p.pcs[PAWN] ^= cb;
            xor_occ(p, them, cb);
            key ^= psq_key(PAWN, them, csq);
        }
    }
    int placed = pt;
    if (promo) placed = promo - 1;  // python-chess type (2..5) -> index (1..4)
    if (pt == KING && (to - from == 2 || from - to == 2)) {
        // castling (stored as the king's two-square move, e.g. e1g1)
        int base = from & 56;
        int rfs = to > from ? base + 7 : base + 0, rts = to > from ? base + 5 : base + 3;
        bb_t rf = bit(rfs), rt = bit(rts);
        p.pcs[ROOK] ^= rf | rt;
        xor_occ(p, us, rf | rt);
        key ^= psq_key(ROOK, us, rfs) ^ psq_key(ROOK, us, rts);
    }
    xor_pcs(p, placed, tb);  // the target square is empty at this point (a captured piece was removed above)
    xor_occ(p, us, tb);
    key ^= psq_key(placed, us, to);
    p.turn = (uint8_t)them;
    p.flags = irrev ? F_IRREV : 0;
    p.key = key ^ state_key(them, p.castling, has_legal_ep(p) ? p.ep : -1);
}

// The board half of make_move alone: pieces, occupancy, castling rights, ep square, clocks, side to move.  key and flags are left
// zero: the training-tensor encoder walks a game with this (one wave per game, a chain of dependent steps) and computes the keys,
// the irreversibility flag and the repetition flags of all plies in parallel afterwards (mcts_kernels.hpp: k_replay_raw,
// k_ply_keys, k_ply_rep).  Must stay in step with make_move above (tests/test_gpu_parity.py::test_encode_steps_* compare the result
// with the oracle).
SC_HD void make_move_board(Position& p, move_t m) {
    int from = mv_from(m), to = mv_to(m), promo = mv_promo(m);
    int us = p.turn, them = !us;
    bool zero = is_zeroing(p, m);
    int old_ep = p.ep;
    p.ep = -1;
    p.halfmove = zero ? 0 : (uint16_t)(p.halfmove + 1);
    if (us == BLACK) p.fullmove++;
    bb_t fb = bit(from), tb = bit(to);
    int pt = piece_type_at(p, from);
    int cap = (occ_c(p, them) & tb) ? piece_type_at(p, to) : -1;
    xor_pcs(p, pt, fb);
    xor_occ(p, us, fb);
    p.castling &= (uint8_t)~(castle_bit_for_sq(from) | castle_bit_for_sq(to));
    if (pt == KING) p.castling &= us ? (uint8_t)~3 : (uint8_t)~12;
    if (cap >= 0) {
        xor_pcs(p, cap, tb);
        xor_occ(p, them, tb);
    }
    if (pt == PAWN) {
        int diff = to - from;
        if (diff == 16 && (from >> 3) == 1) p.ep = (int8_t)(from + 8);
        else if (diff == -16 && (from >> 3) == 6) p.ep = (int8_t)(from - 8);
        else if (to == old_ep && (diff == 7 || diff == 9 || diff == -7 || diff == -9) && cap < 0) {
            bb_t cb = bit(old_ep + (us ? -8 : 8));
            p.pcs[PAWN] ^= cb;
            xor_occ(p, them, cb);
        }
    }
    int placed = promo ? promo - 1 : pt;
    if (pt == KING && (to - from == 2 || from - to == 2)) {
        int base = from & 56;
        bb_t rf = bit(to > from ? base + 7 : base + 0), rt = bit(to > from ? base + 5 : base + 3);
        p.pcs[ROOK] ^= rf | rt;
        xor_occ(p, us, rf | rt);
    }
    xor_pcs(p, placed, tb);
    xor_occ(p, us, tb);
    p.turn = (uint8_t)them;
    p.flags = 0;
    p.key = 0;
}

// ------------------------------------------------------------------ legal move generation
// python-chess _slider_blockers(king)
SC_HD bb_t slider_blockers(const Position& p, int king) {
    int us = p.turn;
    bb_t rq = p.pcs[ROOK] | p.pcs[QUEEN], bq = p.pcs[BISHOP] | p.pcs[QUEEN];
    bb_t snipers = ((rook_attacks(king, 0) & rq) | (bishop_attacks(king, 0) & bq)) & occ_c(p, !us);
    bb_t occ = all_occ(p), blockers = 0;
    while (snipers) {
        int s = msb(snipers);
        snipers ^= bit(s);
        bb_t b = between(king, s) & occ;
        if (b && (b & (b - 1)) == 0) blockers |= b;
    }
    return blockers & occ_c(p, us);
}

// caller-provided storage (LDS on the device, a local array on the host), capacity MAX_MOVES
struct MoveList {
    move_t* m;
    int n;
};

// squares attacked by `color` when our king is lifted off the board (sliders x-ray through it): the exact
// "king may not go there" set, also used for the castling path tests (python-chess _attacked_for_king)
SC_HD bb_t danger_map(const Position& p, int color, bb_t occ_no_king) {
    bb_t them = occ_c(p, color);
    bb_t d = pawn_attacks_bb(color, p.pcs[PAWN] & them) | knight_attacks_bb(p.pcs[KNIGHT] & them) | king_attacks_bb(p.pcs[KING] & them);
    bb_t diag = (p.pcs[BISHOP] | p.pcs[QUEEN]) & them;
    while (diag) {
        int s = lsb(diag);
        diag &= diag - 1;
        d |= bishop_attacks(s, occ_no_king);
    }
    bb_t orth = (p.pcs[ROOK] | p.pcs[QUEEN]) & them;
    while (orth) {
        int s = lsb(orth);
        orth &= orth - 1;
        d |= rook_attacks(s, occ_no_king);
    }
    return d;
}

struct GenCtx {
    const Position* p;
    int king;        // -1 if none
    bb_t blockers;   // our pieces pinned against our king (python-chess _slider_blockers)
    bb_t danger;     // danger_map of the opponent
    MoveList* out;
};
SC_HD void emit(GenCtx& g, int from, int to, int promo) {
    if (g.out->n < MAX_MOVES) g.out->m[g.out->n++] = mk_move(from, to, promo);
}
// python-chess _is_safe for a non-king, non-ep move, as a mask of allowed targets: a pinned piece stays on the
// line through itself and the king
SC_HD bb_t pin_mask(const GenCtx& g, int from) {
    if (g.king < 0 || !(g.blockers & bit(from))) return BB_ALL;
    return line_through(from, g.king);
}
SC_HD void emit_pawn(GenCtx& g, int from, int to) {
    if ((to >> 3) == 0 || (to >> 3) == 7) {
        emit(g, from, to, 5);
        emit(g, from, to, 4);
        emit(g, from, to, 3);
        emit(g, from, to, 2);
    } else
        emit(g, from, to, 0);
}
SC_HD void gen_ep(GenCtx& g, bb_t from_mask, bb_t to_mask) {
    const Position& p = *g.p;
    if (p.ep < 0 || !(bit(p.ep) & to_mask) || (all_occ(p) & bit(p.ep))) return;
    bb_t c = ep_capturers(p, from_mask);
    while (c) {
        int from = msb(c);
        c ^= bit(from);
        if (ep_is_legal(p, from)) emit(g, from, p.ep, 0);
    }
}
// python-chess generate_castling_moves (standard chess)
SC_HD void gen_castling(GenCtx& g, bb_t from_mask, bb_t to_mask) {
    const Position& p = *g.p;
    int us = p.turn, base = us ? 0 : 56, ksq = base + 4;
    bb_t kbb = bit(ksq);
    if (!(p.pcs[KING] & occ_c(p, us) & kbb & from_mask)) return;
    bb_t occ = all_occ(p);
    uint8_t kbit = us ? 1 : 4, qbit = us ? 2 : 8;
    bb_t rooks = p.pcs[ROOK] & occ_c(p, us);
    if ((p.castling & kbit) && (rooks & bit(base + 7)) && (to_mask & bit(base + 7))) {
        if (!(occ & (bit(base + 5) | bit(base + 6))) && !(g.danger & (kbb | bit(base + 5) | bit(base + 6)))) emit(g, ksq, base + 6, 0);
    }
    if ((p.castling & qbit) && (rooks & bit(base + 0)) && (to_mask & bit(base + 0))) {
        if (!(occ & (bit(base + 1) | bit(base + 2) | bit(base + 3))) && !(g.danger & (kbb | bit(base + 3) | bit(base + 2))))
            emit(g, ksq, base + 2, 0);
    }
}
// python-chess generate_pseudo_legal_moves(from_mask, to_mask) with _is_safe folded in as target masks
SC_HD void gen_pseudo(GenCtx& g, bb_t from_mask, bb_t to_mask) {
    const Position& p = *g.p;
    int us = p.turn;
    bb_t ours = occ_c(p, us), theirs = occ_c(p, !us), occ = ours | theirs;
    bb_t non_pawns = ours & ~p.pcs[PAWN] & from_mask;
    while (non_pawns) {
        int from = msb(non_pawns);
        non_pawns ^= bit(from);
        int t = piece_type_at(p, from);
        bb_t moves = piece_attacks(p, from, t, us, occ) & ~ours & to_mask;
        moves &= (from == g.king) ? ~g.danger : pin_mask(g, from);
        while (moves) {
            int to = msb(moves);
            moves ^= bit(to);
            emit(g, from, to, 0);
        }
    }
    if (from_mask & p.pcs[KING]) gen_castling(g, from_mask, to_mask);
    bb_t pawns = p.pcs[PAWN] & ours & from_mask;
    if (!pawns) return;
    bb_t capturers = pawns;
    while (capturers) {
        int from = msb(capturers);
        capturers ^= bit(from);
        bb_t targets = pawn_attacks_bb(us, bit(from)) & theirs & to_mask & pin_mask(g, from);
        while (targets) {
            int to = msb(targets);
            targets ^= bit(to);
            emit_pawn(g, from, to);
        }
    }
    bb_t single, dbl;
    if (us) {
        single = (pawns << 8) & ~occ;
        dbl = (single << 8) & ~occ & (0x0000000000FF0000ULL | 0x00000000FF000000ULL);
    } else {
        single = (pawns >> 8) & ~occ;
        dbl = (single >> 8) & ~occ & (0x0000FF0000000000ULL | 0x000000FF00000000ULL);
    }
    single &= to_mask;
    dbl &= to_mask;
    // a pinned pawn may only push along the pin line (i.e. a file pin)
    bb_t pinned = g.king >= 0 ? (g.blockers & pawns) : 0;
    while (single) {
        int to = msb(single);
        single ^= bit(to);
        int from = to + (us ? -8 : 8);
        if ((pinned & bit(from)) && !(line_through(from, g.king) & bit(to))) continue;
        emit_pawn(g, from, to);
    }
    while (dbl) {
        int to = msb(dbl);
        dbl ^= bit(to);
        int from = to + (us ? -16 : 16);
        if ((pinned & bit(from)) && !(line_through(from, g.king) & bit(to))) continue;
        emit(g, from, to, 0);
    }
    if (p.ep >= 0) gen_ep(g, from_mask, to_mask);
}

// python-chess generate_legal_moves(); returns whether the side to move is in check
SC_HD bool gen_legal(const Position& p, MoveList& out) {
    out.n = 0;
    GenCtx g;
    g.p = &p;
    g.out = &out;
    int us = p.turn;
    bb_t kbb = p.pcs[KING] & occ_c(p, us);
    if (!kbb) {
        g.king = -1;
        g.blockers = 0;
        g.danger = 0;
        gen_pseudo(g, BB_ALL, BB_ALL);
        return false;
    }
    int king = msb(kbb);
    bb_t occ = all_occ(p);
    g.king = king;
    g.blockers = slider_blockers(p, king);
    g.danger = danger_map(p, !us, occ ^ kbb);
    if (!(g.danger & kbb)) {
        gen_pseudo(g, BB_ALL, BB_ALL);
        return false;
    }
    // _generate_evasions: king steps first, then (single checker) captures / interpositions, then ep
    bb_t checkers = attackers_mask(p, !us, king, occ);
    bb_t kt = king_attacks_bb(kbb) & ~occ_c(p, us) & ~g.danger;
    while (kt) {
        int to = msb(kt);
        kt ^= bit(to);
        emit(g, king, to, 0);
    }
    int checker = msb(checkers);
    if (bit(checker) == checkers) {
        bb_t target = between(king, checker) | checkers;
        gen_pseudo(g, ~p.pcs[KING], target);
        if (p.ep >= 0 && !(bit(p.ep) & target)) {
            int last_double = p.ep + (us ? -8 : 8);
            if (last_double == checker) gen_ep(g, BB_ALL, BB_ALL);
        }
    }
    return true;
}

// python-chess has_insufficient_material(color)
SC_HD bool insufficient_side(const Position& p, int color) {
    bb_t own = occ_c(p, color);
    if (own & (p.pcs[PAWN] | p.pcs[ROOK] | p.pcs[QUEEN])) return false;
    if (own & p.pcs[KNIGHT]) return popcnt(own) <= 2 && !(occ_c(p, !color) & ~p.pcs[KING] & ~p.pcs[QUEEN]);
    if (own & p.pcs[BISHOP]) {
        const bb_t dark = 0xAA55AA55AA55AA55ULL;
        bool same = !(p.pcs[BISHOP] & dark) || !(p.pcs[BISHOP] & ~dark);
        return same && !p.pcs[PAWN] && !p.pcs[KNIGHT];
    }
    return true;
}

// ------------------------------------------------------------------ action index (4672-wide)
// Move::encode after Move::rotate for Black: src/chess.rs:504-551, src/backends/torch.rs:162-171;
// src/queenmoves.rs:3-34, src/knightmoves.rs:7-31, src/underpromotions.rs:6-33.  Returns -1 if unencodable.
SC_HD int move_index(move_t m, int turn) {
    int fr = mv_from(m) >> 3, ff = mv_from(m) & 7, tr = mv_to(m) >> 3, tf = mv_to(m) & 7, promo = mv_promo(m);
    if (turn == BLACK) {
        fr = 7 - fr;
        tr = 7 - tr;
    }
    int d0 = tr - fr, d1 = tf - ff;
    int a0 = d0 < 0 ? -d0 : d0, a1 = d1 < 0 ? -d1 : d1;
    int base = fr * 584 + ff * 73;
    if ((promo == 0 || promo == 5) && (d0 == 0 || d1 == 0 || a0 == a1) && (a0 | a1)) {
        int s0 = (d0 > 0) - (d0 < 0), s1 = (d1 > 0) - (d1 < 0);
        // (1,0)=0 (1,1)=1 (0,1)=2 (-1,1)=3 (-1,0)=4 (-1,-1)=5 (0,-1)=6 (1,-1)=7
        int dir = s0 == 1 ? (s1 == 0 ? 0 : s1 == 1 ? 1 : 7) : s0 == 0 ? (s1 == 1 ? 2 : 6) : (s1 == 1 ? 3 : s1 == 0 ? 4 : 5);
        int dist = a0 > a1 ? a0 : a1;
        return base + dir * 7 + dist - 1;
    }
    if ((a0 == 2 && a1 == 1) || (a0 == 1 && a1 == 2)) {
        // (2,1)=0 (1,2)=1 (-1,2)=2 (-2,1)=3 (-2,-1)=4 (-1,-2)=5 (1,-2)=6 (2,-1)=7
        int k = d0 == 2 ? (d1 == 1 ? 0 : 7) : d0 == 1 ? (d1 == 2 ? 1 : 6) : d0 == -1 ? (d1 == 2 ? 2 : 5) : (d1 == 1 ? 3 : 4);
        return base + 56 + k;
    }
    if (promo >= 2 && promo <= 4 && fr == 6 && tr == 7 && a1 <= 1) return base + 64 + (d1 + 1) * 3 + (promo - 2);
    return -1;
}

// ------------------------------------------------------------------ synthetic evaluator (tests)
// Integer-hash priors/values, bit-identical to oracle/mcts.c:orc_eval_synth so that search parity
// can be asserted exactly (GPU bf16 logits cannot be bit-identical to a CPU network).
SC_HD uint64_t synth_pos_hash(const Position& p) {
    uint64_t h = 0x243F6A8885A308D3ULL;
    bb_t occ = all_occ(p);
    while (occ) {
        int sq = lsb(occ);
        occ &= occ - 1;
        int t = piece_type_at(p, sq);
        uint64_t code = (uint64_t)(t + 1) + (((p.occ[WHITE] >> sq) & 1) ? 0 : 6);
        h = mix64(h ^ ((code << 8) | (uint64_t)sq));
    }
    uint64_t tail = (uint64_t)p.turn | ((uint64_t)p.castling << 1) | ((uint64_t)(p.ep + 1) << 5) | ((uint64_t)p.halfmove << 12);
    return mix64(h ^ tail);
}
SC_HD uint32_t synth_weight(uint64_t h, move_t m) { return 1u + (uint32_t)(mix64(h ^ ((uint64_t)m * 0x9E3779B97F4A7C15ULL)) >> 40); }
SC_HD float synth_value(uint64_t h) { return (float)((int64_t)(mix64(h ^ 0xABCDEFULL) >> 40) - 8388608) / 8388608.0f; }

SC_HD uint64_t sc_rng(uint64_t seed, uint64_t game, uint64_t ply, uint64_t purpose, uint64_t counter) {
    uint64_t h = mix64(seed ^ (game * 0xD1B54A32D192ED03ULL));
    h = mix64(h ^ (ply * 0x8CB92BA72F3D8DD7ULL));
    return mix64(h ^ ((purpose << 48) | counter));
}

}  // namespace sc
