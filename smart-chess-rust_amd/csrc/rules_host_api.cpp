// rules_host_api.cpp -- HOST build of the engine's device rules code (chess_rules.hpp,
// chess_history.hpp) behind a tiny C API, for CPU-only unit tests (perft, move order, repetition,
// outcome, encoder) of the very same source the HIP kernels compile.  Not part of the product path:
// libsc_engine.so runs these functions on the GPU only.
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "chess_history.hpp"

using namespace sc;

struct sct_state {
    std::vector<Position> hist;  // hist[i] = position after i moves
    std::vector<move_t> moves;
    const Position& pos(int i) const { return hist[(size_t)i]; }
};

static void finalize_rep(sct_state* s) {
    int idx = (int)s->hist.size() - 1;
    s->hist[(size_t)idx].flags |= repetition_flags(*s, idx);
}

extern "C" {

sct_state* sct_new() {
    sct_state* s = new sct_state();
    Position p;
    set_startpos(p);
    p.key = position_key(p);
    s->hist.push_back(p);
    return s;
}
void sct_free(sct_state* s) { delete s; }
void sct_reset(sct_state* s) {
    Position p;
    set_startpos(p);
    p.key = position_key(p);
    s->hist.assign(1, p);
    s->moves.clear();
}
int sct_set_fen(sct_state* s, const char* fen) {
    Position p;
    memset(&p, 0, sizeof p);
    int r = 7, f = 0;
    const char* c = fen;
    const char* names = "pnbrqk";
    for (; *c && *c != ' '; c++) {
        if (*c == '/') { r--; f = 0; continue; }
        if (*c >= '1' && *c <= '8') { f += *c - '0'; continue; }
        bool white = *c >= 'A' && *c <= 'Z';
        char lc = (char)(white ? *c + 32 : *c);
        const char* q = strchr(names, lc);
        if (!q || r < 0 || f > 7) return -1;
        int sq = r * 8 + f;
        p.pcs[q - names] |= bit(sq);
        p.occ[white ? WHITE : BLACK] |= bit(sq);
        f++;
    }
    if (*c != ' ') return -1;
    c++;
    p.turn = *c == 'w';
    c++;
    while (*c == ' ') c++;
    for (; *c && *c != ' '; c++) {
        if (*c == 'K') p.castling |= 1;
        if (*c == 'Q') p.castling |= 2;
        if (*c == 'k') p.castling |= 4;
        if (*c == 'q') p.castling |= 8;
    }
    bb_t wk = p.pcs[KING] & p.occ[WHITE], bk = p.pcs[KING] & p.occ[BLACK];
    bb_t wr = p.pcs[ROOK] & p.occ[WHITE], br = p.pcs[ROOK] & p.occ[BLACK];
    if (!(wk & bit(4))) p.castling &= (uint8_t)~3;
    if (!(bk & bit(60))) p.castling &= (uint8_t)~12;
    if (!(wr & bit(7))) p.castling &= (uint8_t)~1;
    if (!(wr & bit(0))) p.castling &= (uint8_t)~2;
    if (!(br & bit(63))) p.castling &= (uint8_t)~4;
    if (!(br & bit(56))) p.castling &= (uint8_t)~8;
    while (*c == ' ') c++;
    p.ep = -1;
    if (*c && *c != '-') {
        p.ep = (int8_t)((c[1] - '1') * 8 + (c[0] - 'a'));
        c += 2;
    } else if (*c)
        c++;
    p.halfmove = 0;
    p.fullmove = 1;
    while (*c == ' ') c++;
    if (*c) {
        p.halfmove = (uint16_t)strtol(c, (char**)&c, 10);
        while (*c == ' ') c++;
        if (*c) p.fullmove = (uint16_t)strtol(c, NULL, 10);
    }
    p.key = position_key(p);
    s->hist.assign(1, p);
    s->moves.clear();
    return 0;
}
int sct_turn(const sct_state* s) { return s->hist.back().turn; }
void sct_push(sct_state* s, uint16_t m) {
    Position p = s->hist.back();
    make_move(p, m);
    s->hist.push_back(p);
    s->moves.push_back(m);
    finalize_rep(s);
}
void sct_pop(sct_state* s) {
    s->hist.pop_back();
    s->moves.pop_back();
}
int sct_legal_moves(const sct_state* s, uint16_t* out, int* in_check) {
    move_t buf[MAX_MOVES];
    MoveList l{buf, 0};
    bool chk = gen_legal(s->hist.back(), l);
    if (in_check) *in_check = chk;
    memcpy(out, l.m, sizeof(move_t) * (size_t)l.n);
    return l.n;
}
static uint64_t perft(const Position& p, int depth) {
    move_t buf[MAX_MOVES];
    MoveList l{buf, 0};
    gen_legal(p, l);
    if (depth == 1) return (uint64_t)l.n;
    uint64_t t = 0;
    for (int i = 0; i < l.n; i++) {
        Position q = p;
        make_move(q, l.m[i]);
        t += perft(q, depth - 1);
    }
    return t;
}
uint64_t sct_perft(const sct_state* s, int depth) { return depth <= 0 ? 1 : perft(s->hist.back(), depth); }
int sct_is_repetition(const sct_state* s, int count) { return is_repetition(*s, (int)s->hist.size() - 1, count); }
int sct_outcome(const sct_state* s, int* winner) { return outcome_claim_draw(*s, (int)s->hist.size() - 1, winner); }
void sct_encode(const sct_state* s, int8_t* boards, int32_t* meta) {
    int idx = (int)s->hist.size() - 1;
    for (int px = 0; px < 64; px++) encode_cell(*s, idx, px, boards + px * 112);
    encode_meta(s->hist.back(), meta);
}
int sct_move_index(uint16_t m, int turn) { return move_index(m, turn); }
uint64_t sct_pos_hash(const sct_state* s) { return synth_pos_hash(s->hist.back()); }
uint64_t sct_key(const sct_state* s) { return s->hist.back().key; }
uint64_t sct_key_full(const sct_state* s) { return position_key(s->hist.back()); }
uint64_t sct_rng(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) { return sc_rng(a, b, c, d, e); }
void sct_synth_eval(const sct_state* s, float* priors, float* value) {
    move_t buf[MAX_MOVES];
    MoveList l{buf, 0};
    gen_legal(s->hist.back(), l);
    uint64_t h = synth_pos_hash(s->hist.back());
    uint64_t sum = 0;
    for (int i = 0; i < l.n; i++) sum += synth_weight(h, l.m[i]);
    for (int i = 0; i < l.n; i++) priors[i] = (float)synth_weight(h, l.m[i]) / (float)sum;
    *value = synth_value(h);
}
}
