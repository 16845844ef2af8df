// chess_history.hpp -- history-dependent rules and the plane encoder, over a "chain" of positions.
//
// A Chain gives random access by ply index to the positions of one line of play (game history
// followed by the tree path): chain.pos(i) is the position after i moves; its .key and .flags
// (F_IRREV = "the move that led here was irreversible") are what repetition detection needs.
//
// Restates python-chess is_repetition / outcome(claim_draw=True) as called from the reference at
// src/chess.rs:374-379 and :719-729, and the reference's own encoder src/chess.rs:593-663, 805-877.
#pragma once
#include "chess_rules.hpp"

namespace sc {

// python-chess Board.is_repetition(count) at chain index idx
template <class Chain>
SC_HD bool is_repetition(const Chain& ch, int idx, int count) {
    bb_t key0 = ch.pos(idx).key;
    int i = idx;
    for (;;) {
        if (count <= 1) return true;
        if (i < count - 1) break;
        bool irrev = (ch.pos(i).flags & F_IRREV) != 0;
        i--;
        if (irrev) break;
        if (ch.pos(i).key == key0) count--;
    }
    return false;
}

// Termination numbering of src/chess.rs:88-99
enum { T_NONE = 0, T_CHECKMATE = 1, T_STALEMATE = 2, T_INSUFFICIENT = 3, T_SEVENTYFIVE = 4, T_FIVEFOLD = 5, T_FIFTY = 6, T_THREEFOLD = 7 };

// python-chess Board.outcome(claim_draw=True) (src/chess.rs:719-729). winner: 1 white, 0 black, -1 none.
template <class Chain>
SC_HD int outcome_claim_draw(const Chain& ch, int idx, int* winner) {
    const Position& p = ch.pos(idx);
    move_t buf1[MAX_MOVES], buf2[MAX_MOVES];
    MoveList l{buf1, 0};
    bool in_check = gen_legal(p, l);
    *winner = -1;
    if (in_check && l.n == 0) {
        *winner = !p.turn;
        return T_CHECKMATE;
    }
    if (insufficient_side(p, WHITE) && insufficient_side(p, BLACK)) return T_INSUFFICIENT;
    if (l.n == 0) return T_STALEMATE;
    if (p.halfmove >= 150) return T_SEVENTYFIVE;
    if (is_repetition(ch, idx, 5)) return T_FIVEFOLD;
    if (p.halfmove >= 100) return T_FIFTY;
    if (p.halfmove >= 99) {
        for (int i = 0; i < l.n; i++) {
            if (is_zeroing(p, l.m[i])) continue;
            Position q = p;
            make_move(q, l.m[i]);
            MoveList l2{buf2, 0};
            gen_legal(q, l2);
            if (q.halfmove >= 100 && l2.n > 0) return T_FIFTY;
        }
    }
    // can_claim_threefold_repetition
    int lo = idx;
    while (lo > 0 && !(ch.pos(lo).flags & F_IRREV)) lo--;
    int cnt = 1;
    for (int i = lo; i < idx; i++)
        if (ch.pos(i).key == p.key) cnt++;
    if (cnt >= 3) return T_THREEFOLD;
    for (int j = 0; j < l.n; j++) {
        Position q = p;
        make_move(q, l.m[j]);
        int c2 = q.key == p.key ? 1 : 0;
        for (int i = lo; i < idx; i++)
            if (ch.pos(i).key == q.key) c2++;
        if (c2 >= 2) return T_THREEFOLD;
    }
    return T_NONE;
}

// One board cell of _encode (src/chess.rs:845-877): the 112 plane bytes of output pixel `px`
// (= rank*8+file in the MOVER's frame) for the position at chain index idx, history newest first,
// every board rotated by the current mover's colour (Board::rotate :594-621, encode_pieces :623-650).
template <class Chain>
SC_HD void encode_cell(const Chain& ch, int idx, int px, int8_t* cell /*112*/) {
    int turn = ch.pos(idx).turn;
    int src = turn == BLACK ? (px ^ 56) : px;  // rank flip, file unchanged (Square::rotate :504-509)
    bb_t sb = bit(src);
    for (int j = 0; j < 8; j++) {
        int8_t* c = cell + 14 * j;
        for (int k = 0; k < 14; k++) c[k] = 0;
        if (j > idx) continue;
        const Position& h = ch.pos(idx - j);
        if ((h.occ[0] | h.occ[1]) & sb) {
            int t = piece_type_at(h, src);
            int is_white = (h.occ[WHITE] & sb) ? 1 : 0;
            int mover_side = turn == BLACK ? !is_white : is_white;  // colours swapped for Black
            c[t + (mover_side ? 0 : 6)] = 1;
        }
        c[12] = (h.flags & F_REP2) ? 1 : 0;
        c[13] = (h.flags & F_REP3) ? 1 : 0;
    }
}
// Board::encode_meta (src/chess.rs:652-662) with the mover-first castling pairs of :380-391
SC_HD void encode_meta(const Position& p, int32_t* meta /*7*/) {
    int t = p.turn;
    uint8_t mk_ = t ? 1 : 4, mq = t ? 2 : 8, ok = t ? 4 : 1, oq = t ? 8 : 2;
    meta[0] = t;
    meta[1] = p.fullmove;
    meta[2] = (p.castling & mk_) != 0;
    meta[3] = (p.castling & mq) != 0;
    meta[4] = (p.castling & ok) != 0;
    meta[5] = (p.castling & oq) != 0;
    meta[6] = p.halfmove;
}

// Fill the history-dependent fields of the position at chain index idx (REP2/REP3); key and F_IRREV
// are already set by make_move.
template <class Chain>
SC_HD uint8_t repetition_flags(const Chain& ch, int idx) {
    uint8_t f = 0;
    if (is_repetition(ch, idx, 2)) f |= F_REP2;
    if (is_repetition(ch, idx, 3)) f |= F_REP3;
    return f;
}

}  // namespace sc
