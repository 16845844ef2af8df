/*
 * oracle/chess.c -- chess rules of the CPU ORACLE (test infrastructure only, see sc_oracle.h).
 *
 * Restates the python-chess 1.11.1 behaviour the reference reaches through pyo3
 * (reference src/chess.rs:356-412 to_board, :665-803 BoardState): generate_legal_moves() order,
 * push(), is_repetition(), outcome(claim_draw=True).  python-chess is absent from the reference
 * tree and from this image, so its published algorithm is restated; the results are pinned by
 * public perft known answers and by the reference's own fixtures (tests/test_oracle_rules.py).
 *
 * Method: mailbox board; pseudo-legal moves are enumerated in python-chess's generation order and
 * filtered by "make the move, is my king attacked?".
 */
#include "sc_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define PT_PAWN 1
#define PT_KNIGHT 2
#define PT_BISHOP 3
#define PT_ROOK 4
#define PT_QUEEN 5
#define PT_KING 6

#define BB(sq) (1ULL << (sq))
#define ALL (~0ULL)

static inline int sq_rank(int sq) { return sq >> 3; }
static inline int sq_file(int sq) { return sq & 7; }
static inline int on_board(int r, int f) { return r >= 0 && r < 8 && f >= 0 && f < 8; }
static inline int color_of(int8_t p) { return p > 0 ? ORC_WHITE : ORC_BLACK; }
static inline int type_of(int8_t p) { return p > 0 ? p : -p; }
static inline orc_move mk(int from, int to, int promo) { return (orc_move)(from | (to << 6) | (promo << 12)); }
static inline int m_from(orc_move m) { return m & 63; }
static inline int m_to(orc_move m) { return (m >> 6) & 63; }
static inline int m_promo(orc_move m) { return (m >> 12) & 7; }

static const int KN_D[8][2] = {{2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1}, {-1, -2}, {1, -2}, {2, -1}};
static const int KG_D[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
static const int ROOK_D[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
static const int BISH_D[4][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}};

/* occupancy helpers */
static uint64_t occ_of(const orc_pos* p, int color) {
    uint64_t b = 0;
    for (int s = 0; s < 64; s++)
        if (p->board[s] && color_of(p->board[s]) == color) b |= BB(s);
    return b;
}
static int king_sq(const orc_pos* p, int color) {
    int8_t k = color ? PT_KING : -PT_KING;
    for (int s = 63; s >= 0; s--)
        if (p->board[s] == k) return s;
    return -1;
}

/* slider attack set from sq over a mailbox with an explicit occupancy mask */
static uint64_t slide(uint64_t occ, int sq, const int d[4][2]) {
    uint64_t a = 0;
    for (int i = 0; i < 4; i++) {
        int r = sq_rank(sq) + d[i][0], f = sq_file(sq) + d[i][1];
        while (on_board(r, f)) {
            int t = r * 8 + f;
            a |= BB(t);
            if (occ & BB(t)) break;
            r += d[i][0];
            f += d[i][1];
        }
    }
    return a;
}
static uint64_t step_attacks(int sq, const int d[8][2]) {
    uint64_t a = 0;
    for (int i = 0; i < 8; i++) {
        int r = sq_rank(sq) + d[i][0], f = sq_file(sq) + d[i][1];
        if (on_board(r, f)) a |= BB(r * 8 + f);
    }
    return a;
}
/* squares a pawn of `color` standing on sq attacks (python-chess BB_PAWN_ATTACKS[color][sq]) */
static uint64_t pawn_attacks(int color, int sq) {
    uint64_t a = 0;
    int r = sq_rank(sq) + (color ? 1 : -1);
    for (int df = -1; df <= 1; df += 2) {
        int f = sq_file(sq) + df;
        if (on_board(r, f)) a |= BB(r * 8 + f);
    }
    return a;
}
static uint64_t all_occ(const orc_pos* p) {
    uint64_t b = 0;
    for (int s = 0; s < 64; s++)
        if (p->board[s]) b |= BB(s);
    return b;
}
/* attack set of the piece standing on sq (python-chess attacks_mask) */
static uint64_t attacks_from(const orc_pos* p, int sq, uint64_t occ) {
    int8_t pc = p->board[sq];
    switch (type_of(pc)) {
        case PT_PAWN: return pawn_attacks(color_of(pc), sq);
        case PT_KNIGHT: return step_attacks(sq, KN_D);
        case PT_KING: return step_attacks(sq, KG_D);
        case PT_BISHOP: return slide(occ, sq, BISH_D);
        case PT_ROOK: return slide(occ, sq, ROOK_D);
        case PT_QUEEN: return slide(occ, sq, BISH_D) | slide(occ, sq, ROOK_D);
    }
    return 0;
}
/* mask of pieces of `by` attacking sq given occupancy occ (python-chess attackers_mask) */
static uint64_t attackers(const orc_pos* p, int by, int sq, uint64_t occ) {
    uint64_t a = 0;
    uint64_t rk = slide(occ, sq, ROOK_D), bs = slide(occ, sq, BISH_D);
    uint64_t kn = step_attacks(sq, KN_D), kg = step_attacks(sq, KG_D);
    uint64_t pw = pawn_attacks(!by, sq); /* squares from which a `by` pawn attacks sq */
    for (int s = 0; s < 64; s++) {
        int8_t pc = p->board[s];
        if (!pc || color_of(pc) != by || !(occ & BB(s))) continue;
        int t = type_of(pc);
        uint64_t b = BB(s);
        if ((t == PT_ROOK || t == PT_QUEEN) && (rk & b)) a |= b;
        if ((t == PT_BISHOP || t == PT_QUEEN) && (bs & b)) a |= b;
        if (t == PT_KNIGHT && (kn & b)) a |= b;
        if (t == PT_KING && (kg & b)) a |= b;
        if (t == PT_PAWN && (pw & b)) a |= b;
    }
    return a;
}
static int attacked(const orc_pos* p, int by, int sq, uint64_t occ) { return attackers(p, by, sq, occ) != 0; }

/* python-chess between(a,b): squares strictly between on a shared line, else 0 */
static uint64_t between(int a, int b) {
    int dr = sq_rank(b) - sq_rank(a), df = sq_file(b) - sq_file(a);
    if (!(dr == 0 || df == 0 || abs(dr) == abs(df)) || a == b) return 0;
    int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
    uint64_t m = 0;
    int r = sq_rank(a) + sr, f = sq_file(a) + sf;
    while (r * 8 + f != b) {
        m |= BB(r * 8 + f);
        r += sr;
        f += sf;
    }
    return m;
}
/* python-chess ray(a,b): whole line through a and b, edge to edge (including a and b), else 0 */
static uint64_t ray(int a, int b) {
    int dr = sq_rank(b) - sq_rank(a), df = sq_file(b) - sq_file(a);
    if (!(dr == 0 || df == 0 || abs(dr) == abs(df)) || a == b) return 0;
    int sr = (dr > 0) - (dr < 0), sf = (df > 0) - (df < 0);
    uint64_t m = BB(a);
    for (int dir = -1; dir <= 1; dir += 2) {
        int r = sq_rank(a) + dir * sr, f = sq_file(a) + dir * sf;
        while (on_board(r, f)) {
            m |= BB(r * 8 + f);
            r += dir * sr;
            f += dir * sf;
        }
    }
    return m;
}

/* ---------------------------------------------------------------- make move (Board.push) */
static int is_zeroing(const orc_pos* p, orc_move m) {
    int f = m_from(m), t = m_to(m);
    if (type_of(p->board[f]) == PT_PAWN || type_of(p->board[t]) == PT_PAWN) return 1; /* touched & pawns */
    if (p->board[t] && color_of(p->board[t]) != p->turn) return 1;                     /* capture */
    if (p->board[f] && color_of(p->board[f]) != p->turn) return 1;
    return 0;
}
static uint8_t castle_bit_for_sq(int sq) {
    switch (sq) {
        case 7: return 1;
        case 0: return 2;
        case 63: return 4;
        case 56: return 8;
    }
    return 0;
}
static void make_move(orc_pos* p, orc_move m) {
    int from = m_from(m), to = m_to(m), promo = m_promo(m);
    int us = p->turn;
    int old_ep = p->ep;
    p->ep = -1;
    p->halfmove += 1;
    if (us == ORC_BLACK) p->fullmove += 1;
    if (is_zeroing(p, m)) p->halfmove = 0;

    int8_t pc = p->board[from];
    int pt = type_of(pc);
    p->board[from] = 0;
    /* castling rights: any touch of a rook square or a king move */
    p->castling &= (uint8_t)~(castle_bit_for_sq(from) | castle_bit_for_sq(to));
    if (pt == PT_KING) p->castling &= us ? (uint8_t)~3 : (uint8_t)~12;

    if (pt == PT_PAWN) {
        int diff = to - from;
        if (diff == 16 && sq_rank(from) == 1) p->ep = (int8_t)(from + 8);
        else if (diff == -16 && sq_rank(from) == 6) p->ep = (int8_t)(from - 8);
        else if (to == old_ep && (abs(diff) == 7 || abs(diff) == 9) && !p->board[to]) {
            int cap = old_ep + (us ? -8 : 8);
            p->board[cap] = 0;
        }
    }
    if (promo) pt = promo;
    /* castling: king moves two files on its back rank (python-chess stores e1g1 in the move stack) */
    if (type_of(pc) == PT_KING && abs(sq_file(to) - sq_file(from)) == 2 && sq_rank(from) == sq_rank(to)) {
        int rank = sq_rank(from) * 8;
        if (sq_file(to) == 6) {
            p->board[rank + 5] = p->board[rank + 7];
            p->board[rank + 7] = 0;
        } else {
            p->board[rank + 3] = p->board[rank + 0];
            p->board[rank + 0] = 0;
        }
    }
    p->board[to] = (int8_t)(us ? pt : -pt);
    p->turn = (uint8_t)!us;
}

/* ---------------------------------------------------------------- move generation */
typedef struct {
    orc_move m[ORC_MAX_MOVES];
    int n;
} mlist;
static void emit(mlist* l, int from, int to, int promo) { l->m[l->n++] = mk(from, to, promo); }
static void emit_pawn(mlist* l, int from, int to) {
    if (sq_rank(to) == 0 || sq_rank(to) == 7) {
        emit(l, from, to, PT_QUEEN);
        emit(l, from, to, PT_ROOK);
        emit(l, from, to, PT_BISHOP);
        emit(l, from, to, PT_KNIGHT);
    } else
        emit(l, from, to, 0);
}

/* generate_pseudo_legal_ep */
static void gen_ep(const orc_pos* p, uint64_t from_mask, uint64_t to_mask, mlist* l) {
    if (p->ep < 0 || !(BB(p->ep) & to_mask)) return;
    if (p->board[p->ep]) return;
    int us = p->turn;
    int rank = us ? 4 : 3;
    uint64_t cand = pawn_attacks(!us, p->ep); /* squares from which our pawn attacks ep */
    for (int s = 63; s >= 0; s--) {
        if (!(cand & BB(s)) || !(from_mask & BB(s))) continue;
        if (sq_rank(s) != rank) continue;
        if (p->board[s] != (us ? PT_PAWN : -PT_PAWN)) continue;
        emit(l, s, p->ep, 0);
    }
}

/* generate_castling_moves (standard chess) */
static void gen_castling(const orc_pos* p, uint64_t from_mask, uint64_t to_mask, mlist* l) {
    int us = p->turn;
    int base = us ? 0 : 56;
    int ksq = base + 4;
    if (p->board[ksq] != (us ? PT_KING : -PT_KING) || !(from_mask & BB(ksq))) return;
    uint64_t occ = all_occ(p);
    uint64_t occ_nk = occ & ~BB(ksq);
    /* scan_reversed over the rook squares: h-side first, then a-side */
    uint8_t kbit = us ? 1 : 4, qbit = us ? 2 : 8;
    int8_t rook = us ? PT_ROOK : -PT_ROOK;
    if ((p->castling & kbit) && p->board[base + 7] == rook && (to_mask & BB(base + 7))) {
        if (!p->board[base + 5] && !p->board[base + 6] && !attacked(p, !us, ksq, occ_nk) &&
            !attacked(p, !us, base + 5, occ_nk) && !attacked(p, !us, base + 6, occ_nk))
            emit(l, ksq, base + 6, 0);
    }
    if ((p->castling & qbit) && p->board[base + 0] == rook && (to_mask & BB(base + 0))) {
        if (!p->board[base + 1] && !p->board[base + 2] && !p->board[base + 3] && !attacked(p, !us, ksq, occ_nk) &&
            !attacked(p, !us, base + 3, occ_nk) && !attacked(p, !us, base + 2, occ_nk))
            emit(l, ksq, base + 2, 0);
    }
}

/* generate_pseudo_legal_moves(from_mask, to_mask) in python-chess order */
static void gen_pseudo(const orc_pos* p, uint64_t from_mask, uint64_t to_mask, mlist* l) {
    int us = p->turn;
    uint64_t ours = occ_of(p, us), theirs = occ_of(p, !us), occ = ours | theirs;
    uint64_t kings = 0, pawns = 0;
    for (int s = 0; s < 64; s++) {
        if (type_of(p->board[s]) == PT_KING) kings |= BB(s);
        if (p->board[s] == (us ? PT_PAWN : -PT_PAWN)) pawns |= BB(s);
    }
    /* piece moves */
    for (int from = 63; from >= 0; from--) {
        if (!(ours & BB(from)) || !(from_mask & BB(from)) || type_of(p->board[from]) == PT_PAWN) continue;
        uint64_t t = attacks_from(p, from, occ) & ~ours & to_mask;
        for (int to = 63; to >= 0; to--)
            if (t & BB(to)) emit(l, from, to, 0);
    }
    /* castling */
    if (from_mask & kings) gen_castling(p, from_mask, to_mask, l);
    /* pawns */
    pawns &= from_mask;
    if (!pawns) return;
    for (int from = 63; from >= 0; from--) {
        if (!(pawns & BB(from))) continue;
        uint64_t t = pawn_attacks(us, from) & theirs & to_mask;
        for (int to = 63; to >= 0; to--)
            if (t & BB(to)) emit_pawn(l, from, to);
    }
    uint64_t single, dbl;
    if (us) {
        single = (pawns << 8) & ~occ;
        dbl = (single << 8) & ~occ & (0x0000000000FF0000ULL | 0x00000000FF000000ULL);
    } else {
        single = (pawns >> 8) & ~occ;
        dbl = (single >> 8) & ~occ & (0x0000FF0000000000ULL | 0x000000FF00000000ULL);
    }
    single &= to_mask;
    dbl &= to_mask;
    for (int to = 63; to >= 0; to--)
        if (single & BB(to)) emit_pawn(l, to + (us ? -8 : 8), to);
    for (int to = 63; to >= 0; to--)
        if (dbl & BB(to)) emit(l, to + (us ? -16 : 16), to, 0);
    if (p->ep >= 0) gen_ep(p, from_mask, to_mask, l);
}

static int leaves_king_safe(const orc_pos* p, orc_move m) {
    orc_pos q = *p;
    make_move(&q, m);
    int k = king_sq(&q, p->turn);
    if (k < 0) return 1;
    return !attacked(&q, !p->turn, k, all_occ(&q));
}

static void gen_legal(const orc_pos* p, mlist* out) {
    mlist l;
    l.n = 0;
    int us = p->turn;
    int k = king_sq(p, us);
    uint64_t occ = all_occ(p);
    uint64_t checkers = k >= 0 ? attackers(p, !us, k, occ) : 0;
    if (k >= 0 && checkers) {
        /* _generate_evasions */
        uint64_t att = 0;
        for (int s = 63; s >= 0; s--) {
            if (!(checkers & BB(s))) continue;
            int t = type_of(p->board[s]);
            if (t == PT_BISHOP || t == PT_ROOK || t == PT_QUEEN) att |= ray(k, s) & ~BB(s);
        }
        uint64_t ours = occ_of(p, us);
        uint64_t kt = step_attacks(k, KG_D) & ~ours & ~att;
        for (int to = 63; to >= 0; to--)
            if (kt & BB(to)) emit(&l, k, to, 0);
        int checker = 63;
        while (!(checkers & BB(checker))) checker--;
        if (BB(checker) == checkers) {
            uint64_t target = between(k, checker) | checkers;
            uint64_t kings = 0;
            for (int s = 0; s < 64; s++)
                if (type_of(p->board[s]) == PT_KING) kings |= BB(s);
            gen_pseudo(p, ~kings, target, &l);
            if (p->ep >= 0 && !(BB(p->ep) & target)) {
                int last_double = p->ep + (us ? -8 : 8);
                if (last_double == checker) gen_ep(p, ALL, ALL, &l);
            }
        }
    } else {
        gen_pseudo(p, ALL, ALL, &l);
    }
    out->n = 0;
    for (int i = 0; i < l.n; i++)
        if (leaves_king_safe(p, l.m[i])) out->m[out->n++] = l.m[i];
}

static int has_legal_ep(const orc_pos* p) {
    if (p->ep < 0) return 0;
    mlist l;
    l.n = 0;
    gen_ep(p, ALL, ALL, &l);
    for (int i = 0; i < l.n; i++)
        if (leaves_king_safe(p, l.m[i])) return 1;
    return 0;
}

/* ---------------------------------------------------------------- repetition / outcome */
/* _transposition_key equality */
static int same_key(const orc_pos* a, int a_ep_legal, const orc_pos* b, int b_ep_legal) {
    if (memcmp(a->board, b->board, 64)) return 0;
    if (a->turn != b->turn || a->castling != b->castling) return 0;
    int ea = a_ep_legal ? a->ep : -1, eb = b_ep_legal ? b->ep : -1;
    return ea == eb;
}
static int reduces_castling(const orc_pos* p, orc_move m) {
    int f = m_from(m), t = m_to(m);
    uint8_t cr = p->castling;
    if (cr & (castle_bit_for_sq(f) | castle_bit_for_sq(t))) return 1;
    if ((cr & 3) && (p->board[f] == PT_KING || p->board[t] == PT_KING)) return 1;
    if ((cr & 12) && (p->board[f] == -PT_KING || p->board[t] == -PT_KING)) return 1;
    return 0;
}
static int is_irreversible(const orc_pos* before, orc_move m) {
    return is_zeroing(before, m) || reduces_castling(before, m) || has_legal_ep(before);
}
/* position after `idx` moves of the state */
static const orc_pos* pos_at(const orc_state* s, int idx) { return idx == s->n ? &s->cur : &s->stack[idx]; }

/* is_repetition(count) evaluated on the state truncated to `upto` moves */
static int is_repetition_at(const orc_state* s, int upto, int count) {
    const orc_pos* cur = pos_at(s, upto);
    int cur_ep = has_legal_ep(cur);
    int i = upto;
    for (;;) {
        if (count <= 1) return 1;
        if (i < count - 1) break;
        const orc_pos* before = pos_at(s, i - 1);
        orc_move m = s->moves[i - 1];
        i--;
        if (is_irreversible(before, m)) break;
        if (same_key(before, has_legal_ep(before), cur, cur_ep)) count--;
    }
    return 0;
}

static int insufficient_side(const orc_pos* p, int color) {
    int n_own = 0, own_knights = 0, own_bishops = 0, own_prq = 0;
    int opp_non_kq = 0;
    int bishops_dark = 0, bishops_light = 0, pawns_any = 0, knights_any = 0;
    for (int s = 0; s < 64; s++) {
        int8_t pc = p->board[s];
        if (!pc) continue;
        int t = type_of(pc), c = color_of(pc);
        if (t == PT_PAWN) pawns_any = 1;
        if (t == PT_KNIGHT) knights_any = 1;
        if (t == PT_BISHOP) {
            if ((sq_rank(s) + sq_file(s)) & 1) bishops_light = 1;
            else bishops_dark = 1;
        }
        if (c == color) {
            n_own++;
            if (t == PT_KNIGHT) own_knights = 1;
            if (t == PT_BISHOP) own_bishops = 1;
            if (t == PT_PAWN || t == PT_ROOK || t == PT_QUEEN) own_prq = 1;
        } else if (t != PT_KING && t != PT_QUEEN)
            opp_non_kq = 1;
    }
    if (own_prq) return 0;
    if (own_knights) return n_own <= 2 && !opp_non_kq;
    if (own_bishops) {
        int same_color = !bishops_dark || !bishops_light;
        return same_color && !pawns_any && !knights_any;
    }
    return 1;
}

int orc_outcome(orc_state* s, int* termination, int* winner) {
    const orc_pos* p = &s->cur;
    mlist l;
    gen_legal(p, &l);
    int k = king_sq(p, p->turn);
    int in_check = k >= 0 && attacked(p, !p->turn, k, all_occ(p));
    *winner = -1;
    if (in_check && l.n == 0) { *termination = 1; *winner = !p->turn; return 1; }
    if (insufficient_side(p, ORC_WHITE) && insufficient_side(p, ORC_BLACK)) { *termination = 3; return 1; }
    if (l.n == 0) { *termination = 2; return 1; }
    if (p->halfmove >= 150) { *termination = 4; return 1; }
    if (is_repetition_at(s, s->n, 5)) { *termination = 5; return 1; }
    /* claim_draw=True: can_claim_fifty_moves */
    if (p->halfmove >= 100) { *termination = 6; return 1; }
    if (p->halfmove >= 99) {
        for (int i = 0; i < l.n; i++) {
            if (is_zeroing(p, l.m[i])) continue;
            orc_pos q = *p;
            make_move(&q, l.m[i]);
            mlist l2;
            gen_legal(&q, &l2);
            if (q.halfmove >= 100 && l2.n > 0) { *termination = 6; return 1; }
        }
    }
    /* can_claim_threefold_repetition */
    {
        /* collect the reversible window of earlier positions */
        int lo = s->n;
        while (lo > 0) {
            const orc_pos* before = pos_at(s, lo - 1);
            if (is_irreversible(before, s->moves[lo - 1])) break;
            lo--;
        }
        int cur_ep = has_legal_ep(p);
        int cnt = 1;
        for (int i = lo; i < s->n; i++) {
            const orc_pos* q = pos_at(s, i);
            if (same_key(q, has_legal_ep(q), p, cur_ep)) cnt++;
        }
        if (cnt >= 3) { *termination = 7; return 1; }
        for (int j = 0; j < l.n; j++) {
            orc_pos q = *p;
            make_move(&q, l.m[j]);
            int q_ep = has_legal_ep(&q);
            int c2 = 0;
            if (same_key(&q, q_ep, p, cur_ep)) c2++;
            for (int i = lo; i < s->n; i++) {
                const orc_pos* r = pos_at(s, i);
                if (same_key(r, has_legal_ep(r), &q, q_ep)) c2++;
            }
            if (c2 >= 2) { *termination = 7; return 1; }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- public state API */
static void pos_start(orc_pos* p) {
    static const int8_t back[8] = {PT_ROOK, PT_KNIGHT, PT_BISHOP, PT_QUEEN, PT_KING, PT_BISHOP, PT_KNIGHT, PT_ROOK};
    memset(p, 0, sizeof *p);
    for (int f = 0; f < 8; f++) {
        p->board[f] = back[f];
        p->board[8 + f] = PT_PAWN;
        p->board[48 + f] = -PT_PAWN;
        p->board[56 + f] = (int8_t)-back[f];
    }
    p->turn = ORC_WHITE;
    p->castling = 15;
    p->ep = -1;
    p->halfmove = 0;
    p->fullmove = 1;
}
orc_state* orc_state_new(void) {
    orc_state* s = (orc_state*)malloc(sizeof(orc_state));
    orc_state_reset(s);
    return s;
}
void orc_state_free(orc_state* s) { free(s); }
void orc_state_reset(orc_state* s) {
    s->n = 0;
    pos_start(&s->cur);
}
void orc_state_copy(orc_state* dst, const orc_state* src) {
    dst->cur = src->cur;
    dst->n = src->n;
    memcpy(dst->stack, src->stack, sizeof(orc_pos) * (size_t)src->n);
    memcpy(dst->moves, src->moves, sizeof(orc_move) * (size_t)src->n);
}
int orc_turn(const orc_state* s) { return s->cur.turn; }
int orc_ply(const orc_state* s) { return s->n; }
int orc_piece_at(const orc_state* s, int sq) { return s->cur.board[sq]; }
void orc_push(orc_state* s, orc_move m) {
    s->stack[s->n] = s->cur;
    s->moves[s->n] = m;
    s->n++;
    make_move(&s->cur, m);
}
orc_move orc_pop(orc_state* s) {
    s->n--;
    s->cur = s->stack[s->n];
    return s->moves[s->n];
}
int orc_legal_moves(const orc_state* s, orc_move* out) {
    mlist l;
    gen_legal(&s->cur, &l);
    memcpy(out, l.m, sizeof(orc_move) * (size_t)l.n);
    return l.n;
}
int orc_is_check(const orc_state* s) {
    int k = king_sq(&s->cur, s->cur.turn);
    return k >= 0 && attacked(&s->cur, !s->cur.turn, k, all_occ(&s->cur));
}
static uint64_t perft_pos(const orc_pos* p, int depth) {
    mlist l;
    gen_legal(p, &l);
    if (depth == 1) return (uint64_t)l.n;
    uint64_t t = 0;
    for (int i = 0; i < l.n; i++) {
        orc_pos q = *p;
        make_move(&q, l.m[i]);
        t += perft_pos(&q, depth - 1);
    }
    return t;
}
uint64_t orc_perft(orc_state* s, int depth) { return depth <= 0 ? 1 : perft_pos(&s->cur, depth); }
int orc_is_repetition(const orc_state* s, int count) { return is_repetition_at(s, s->n, count); }

static const char PCH[] = " pnbrqk";
int orc_state_set_fen(orc_state* s, const char* fen) {
    orc_pos* p = &s->cur;
    memset(p, 0, sizeof *p);
    s->n = 0;
    int r = 7, f = 0;
    const char* c = fen;
    for (; *c && *c != ' '; c++) {
        if (*c == '/') { r--; f = 0; continue; }
        if (*c >= '1' && *c <= '8') { f += *c - '0'; continue; }
        int white = (*c >= 'A' && *c <= 'Z');
        char lc = (char)(white ? *c + 32 : *c);
        const char* q = strchr(PCH + 1, lc);
        if (!q || r < 0 || f > 7) return -1;
        int pt = (int)(q - PCH);
        p->board[r * 8 + f] = (int8_t)(white ? pt : -pt);
        f++;
    }
    if (*c != ' ') return -1;
    c++;
    p->turn = (*c == 'w');
    c++;
    while (*c == ' ') c++;
    p->castling = 0;
    for (; *c && *c != ' '; c++) {
        if (*c == 'K') p->castling |= 1;
        if (*c == 'Q') p->castling |= 2;
        if (*c == 'k') p->castling |= 4;
        if (*c == 'q') p->castling |= 8;
    }
    /* clean_castling_rights */
    if (p->board[4] != PT_KING) p->castling &= (uint8_t)~3;
    if (p->board[60] != -PT_KING) p->castling &= (uint8_t)~12;
    if (p->board[7] != PT_ROOK) p->castling &= (uint8_t)~1;
    if (p->board[0] != PT_ROOK) p->castling &= (uint8_t)~2;
    if (p->board[63] != -PT_ROOK) p->castling &= (uint8_t)~4;
    if (p->board[56] != -PT_ROOK) p->castling &= (uint8_t)~8;
    while (*c == ' ') c++;
    p->ep = -1;
    if (*c && *c != '-') {
        p->ep = (int8_t)((c[1] - '1') * 8 + (c[0] - 'a'));
        c += 2;
    } else if (*c)
        c++;
    p->halfmove = 0;
    p->fullmove = 1;
    while (*c == ' ') c++;
    if (*c) {
        p->halfmove = (int32_t)strtol(c, (char**)&c, 10);
        while (*c == ' ') c++;
        if (*c) p->fullmove = (int32_t)strtol(c, NULL, 10);
    }
    return 0;
}
int orc_fen(const orc_state* s, char* buf, int cap) {
    const orc_pos* p = &s->cur;
    char tmp[128];
    int n = 0;
    for (int r = 7; r >= 0; r--) {
        int e = 0;
        for (int f = 0; f < 8; f++) {
            int8_t pc = p->board[r * 8 + f];
            if (!pc) { e++; continue; }
            if (e) { tmp[n++] = (char)('0' + e); e = 0; }
            char ch = PCH[type_of(pc)];
            tmp[n++] = (char)(pc > 0 ? ch - 32 : ch);
        }
        if (e) tmp[n++] = (char)('0' + e);
        if (r) tmp[n++] = '/';
    }
    tmp[n++] = ' ';
    tmp[n++] = p->turn ? 'w' : 'b';
    tmp[n++] = ' ';
    if (!p->castling) tmp[n++] = '-';
    if (p->castling & 1) tmp[n++] = 'K';
    if (p->castling & 2) tmp[n++] = 'Q';
    if (p->castling & 4) tmp[n++] = 'k';
    if (p->castling & 8) tmp[n++] = 'q';
    tmp[n++] = ' ';
    if (has_legal_ep(p)) {
        tmp[n++] = (char)('a' + sq_file(p->ep));
        tmp[n++] = (char)('1' + sq_rank(p->ep));
    } else
        tmp[n++] = '-';
    tmp[n] = 0;
    return snprintf(buf, (size_t)cap, "%s %d %d", tmp, p->halfmove, p->fullmove);
}
int orc_move_uci(orc_move m, char* buf) {
    int f = m_from(m), t = m_to(m), pr = m_promo(m);
    int n = 0;
    buf[n++] = (char)('a' + sq_file(f));
    buf[n++] = (char)('1' + sq_rank(f));
    buf[n++] = (char)('a' + sq_file(t));
    buf[n++] = (char)('1' + sq_rank(t));
    if (pr) buf[n++] = PCH[pr];
    buf[n] = 0;
    return n;
}
orc_move orc_move_from_uci(const char* u) {
    int f = (u[1] - '1') * 8 + (u[0] - 'a'), t = (u[3] - '1') * 8 + (u[2] - 'a');
    int pr = 0;
    if (u[4]) {
        const char* q = strchr(PCH + 1, u[4]);
        pr = q ? (int)(q - PCH) : 0;
    }
    return mk(f, t, pr);
}

/* ---------------------------------------------------------------- encoders */
/* Move::encode (src/chess.rs:544-550) after Move::rotate for Black (src/backends/torch.rs:162-171) */
int orc_move_index(orc_move m, int turn) {
    int fr = sq_rank(m_from(m)), ff = sq_file(m_from(m));
    int tr = sq_rank(m_to(m)), tf = sq_file(m_to(m));
    int promo = m_promo(m);
    if (turn == ORC_BLACK) { /* Square::rotate src/chess.rs:504-509 */
        fr = 7 - fr;
        tr = 7 - tr;
    }
    int d0 = tr - fr, d1 = tf - ff;
    /* queenmoves::encode src/queenmoves.rs:3-34 */
    if ((promo == 0 || promo == PT_QUEEN) && (d0 == 0 || d1 == 0 || abs(d0) == abs(d1))) {
        int dist = abs(d0) > abs(d1) ? abs(d0) : abs(d1);
        int s0 = (d0 > 0) - (d0 < 0), s1 = (d1 > 0) - (d1 < 0);
        int dir = -1;
        if (s0 == -1 && s1 == -1) dir = 5;
        else if (s0 == -1 && s1 == 0) dir = 4;
        else if (s0 == -1 && s1 == 1) dir = 3;
        else if (s0 == 0 && s1 == -1) dir = 6;
        else if (s0 == 0 && s1 == 1) dir = 2;
        else if (s0 == 1 && s1 == -1) dir = 7;
        else if (s0 == 1 && s1 == 0) dir = 0;
        else if (s0 == 1 && s1 == 1) dir = 1;
        if (dir >= 0) return fr * 8 * 73 + ff * 73 + dir * 7 + (dist - 1);
    }
    /* knightmoves::encode src/knightmoves.rs:7-31 */
    for (int i = 0; i < 8; i++)
        if (KN_D[i][0] == d0 && KN_D[i][1] == d1) return fr * 8 * 73 + ff * 73 + 56 + i;
    /* underpromotions::encode src/underpromotions.rs:6-33 */
    if ((promo == PT_KNIGHT || promo == PT_BISHOP || promo == PT_ROOK) && fr == 6 && tr == 7 && d1 >= -1 && d1 <= 1) {
        int pidx = promo == PT_KNIGHT ? 0 : promo == PT_BISHOP ? 1 : 2;
        return fr * 8 * 73 + ff * 73 + 64 + (d1 + 1) * 3 + pidx;
    }
    return -1;
}

/* _encode (src/chess.rs:845-877): newest board first, up to 8 boards back to the game root, every
 * board rotated by the CURRENT mover's colour (Board::rotate :594-621, encode_pieces :623-650);
 * meta from the current board un-rotated (encode_meta :652-662, castling pairs :380-391). */
void orc_encode(const orc_state* s, int8_t* boards, int32_t* meta) {
    memset(boards, 0, 8 * 8 * 112);
    int turn = s->cur.turn;
    for (int j = 0; j < 8 && j <= s->n; j++) {
        int idx = s->n - j;
        const orc_pos* p = pos_at(s, idx);
        int rep2 = is_repetition_at(s, idx, 2), rep3 = is_repetition_at(s, idx, 3);
        for (int sq = 0; sq < 64; sq++) {
            int r = sq_rank(sq), f = sq_file(sq);
            int8_t pc = p->board[sq];
            int rr = r;
            if (turn == ORC_BLACK) {
                rr = 7 - r;
                pc = (int8_t)-pc;
            }
            int8_t* cell = boards + (rr * 8 + f) * 112 + 14 * j;
            if (pc) cell[(type_of(pc) - 1) + (pc > 0 ? 0 : 6)] = 1;
        }
        for (int sq = 0; sq < 64; sq++) {
            boards[sq * 112 + 14 * j + 12] = (int8_t)rep2;
            boards[sq * 112 + 14 * j + 13] = (int8_t)rep3;
        }
    }
    const orc_pos* c = &s->cur;
    uint8_t mk_ = turn ? 1 : 4, mq = turn ? 2 : 8, ok = turn ? 4 : 1, oq = turn ? 8 : 2;
    meta[0] = turn;
    meta[1] = c->fullmove;
    meta[2] = (c->castling & mk_) != 0;
    meta[3] = (c->castling & mq) != 0;
    meta[4] = (c->castling & ok) != 0;
    meta[5] = (c->castling & oq) != 0;
    meta[6] = c->halfmove;
}

/* ------------------------------------------------------------------ training-tensor encoder
 * libsmartchess.chess_encode_steps (reference src/lib.rs:46-128), restated LITERALLY: a `Board` snapshot per ply
 * (FromPyObject src/chess.rs:355-412), optionally Board::rotate()d when apply_mirror (src/chess.rs:594-621),
 * pushed to the FRONT of a LOOKBACK-deep deque (BoardHistory::push_front :813-818) and viewed with every stored
 * board rotated once more when the pushed board's turn is Black (BoardHistory::view :827-842). */
typedef struct {
    int8_t pc[64];      /* piece_map: signed piece codes, square = rank*8+file of THIS board */
    int turn;           /* 1 white */
    int rep2, rep3;
    int halfmove, fullmove;
    int ks[2], qs[2];   /* has_{king,queen}side_castling_rights: (.0 = board.turn, .1 = !turn) at extraction */
} tboard;

static void tboard_from_state(const orc_state* s, tboard* b) {
    const orc_pos* c = &s->cur;
    memcpy(b->pc, c->board, 64);
    b->turn = c->turn;
    b->rep2 = is_repetition_at(s, s->n, 2);
    b->rep3 = is_repetition_at(s, s->n, 3);
    b->halfmove = c->halfmove;
    b->fullmove = c->fullmove;
    int t = c->turn;
    uint8_t mk_ = t ? 1 : 4, mq = t ? 2 : 8, ok = t ? 4 : 1, oq = t ? 8 : 2;
    b->ks[0] = (c->castling & mk_) != 0;
    b->ks[1] = (c->castling & ok) != 0;
    b->qs[0] = (c->castling & mq) != 0;
    b->qs[1] = (c->castling & oq) != 0;
}
/* Board::rotate: squares rank -> 7-rank (Square::rotate :504-509), colours swapped, turn flipped, fullmove + 1 if
 * White was to move, castling tuples swapped */
static void tboard_rotate(const tboard* a, tboard* r) {
    for (int sq = 0; sq < 64; sq++) r->pc[sq] = 0;
    for (int sq = 0; sq < 64; sq++)
        if (a->pc[sq]) r->pc[(7 - (sq >> 3)) * 8 + (sq & 7)] = (int8_t)-a->pc[sq];
    r->turn = !a->turn;
    r->rep2 = a->rep2;
    r->rep3 = a->rep3;
    r->halfmove = a->halfmove;
    r->fullmove = a->fullmove + (a->turn == ORC_WHITE ? 1 : 0);
    r->ks[0] = a->ks[1];
    r->ks[1] = a->ks[0];
    r->qs[0] = a->qs[1];
    r->qs[1] = a->qs[0];
}
/* Board::encode_pieces :623-650 into the 14 planes at `base` of a [8][8][112] array */
static void tboard_encode_pieces(const tboard* b, int8_t* full, int base) {
    for (int sq = 0; sq < 64; sq++) {
        int8_t* cell = full + sq * 112 + base;
        for (int k = 0; k < 14; k++) cell[k] = 0;
        int8_t pc = b->pc[sq];
        if (pc) cell[(type_of(pc) - 1) + (pc > 0 ? 0 : 6)] = 1;
        cell[12] = (int8_t)b->rep2;
        cell[13] = (int8_t)b->rep3;
    }
}

/* One game.  n_steps plies; next[i] = move played at ply i; child moves/counts of ply i at [coff[i], coff[i+1]).
 * Outputs per ply: boards int8[7168], meta int32[7], dist float[4672], idx int32[224] (+ n_idx).
 * Returns 0, or -(i+1) when ply i's next move is not legal, or 1000+i when its children are not exactly the legal
 * moves (the reference panics: "inconsistent moves", lib.rs:64-76). */
int orc_encode_steps(int n_steps, const orc_move* next, const orc_move* cmv, const uint32_t* ccnt, const uint32_t* coff,
                     int apply_mirror, int8_t* boards, int32_t* meta, float* dist, int32_t* idx, int32_t* n_idx) {
    orc_state* st = orc_state_new();
    tboard hist[8];
    int nh = 0;
    int rc = 0;
    for (int i = 0; i < n_steps; i++) {
        orc_move legal[ORC_MAX_MOVES];
        int nl = orc_legal_moves(st, legal);
        int nc = (int)(coff[i + 1] - coff[i]);
        const orc_move* cm = cmv + coff[i];
        const uint32_t* cc = ccnt + coff[i];
        /* sets must be equal (symmetric difference empty) */
        int bad = 0;
        for (int a = 0; a < nc && !bad; a++) {
            int f = 0;
            for (int b = 0; b < nl; b++) f |= legal[b] == cm[a];
            if (!f) bad = 1;
        }
        for (int b = 0; b < nl && !bad; b++) {
            int f = 0;
            for (int a = 0; a < nc; a++) f |= legal[b] == cm[a];
            if (!f) bad = 1;
        }
        if (bad) {
            rc = 1000 + i;
            break;
        }
        int has_next = 0;
        for (int b = 0; b < nl; b++) has_next |= legal[b] == next[i];
        if (!has_next) {
            rc = -(i + 1);
            break;
        }
        tboard tb, step;
        tboard_from_state(st, &tb);
        if (apply_mirror) tboard_rotate(&tb, &step);
        else step = tb;
        /* rotate_and_encode: by the ORIGINAL mover (lib.rs:85-92) */
        int ori_turn = apply_mirror ? !step.turn : step.turn;
        /* push_front */
        if (nh == 8) nh = 7;
        for (int k = nh; k > 0; k--) hist[k] = hist[k - 1];
        hist[0] = step;
        nh++;
        int8_t* B = boards + (size_t)i * 7168;
        memset(B, 0, 7168);
        int rot = step.turn == ORC_BLACK;
        for (int k = 0; k < nh; k++) {
            tboard v;
            if (rot) tboard_rotate(&hist[k], &v);
            else v = hist[k];
            tboard_encode_pieces(&v, B, 14 * k);
        }
        int32_t* M = meta + (size_t)i * 7;
        M[0] = step.turn;
        M[1] = step.fullmove;
        M[2] = step.ks[0];
        M[3] = step.qs[0];
        M[4] = step.ks[1];
        M[5] = step.qs[1];
        M[6] = step.halfmove;
        for (int b = 0; b < nl; b++) idx[(size_t)i * 224 + b] = orc_move_index(legal[b], ori_turn);
        n_idx[i] = nl;
        float* D = dist + (size_t)i * 4672;
        for (int k = 0; k < 4672; k++) D[k] = 0.f;
        uint32_t sum = 0;
        for (int a = 0; a < nc; a++) sum += cc[a];
        for (int a = 0; a < nc; a++) {
            int ind = orc_move_index(cm[a], ori_turn);
            D[ind] = (float)cc[a] / ((float)sum + 1e-5f);
        }
        orc_push(st, next[i]);
    }
    orc_state_free(st);
    return rc;
}
