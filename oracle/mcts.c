/*
 * oracle/mcts.c -- PUCT search, predict contract and self-play driver of the CPU ORACLE
 * (test infrastructure only, see sc_oracle.h).
 *
 * Restates, line by line where arithmetic matters:
 *   uct            src/mcts.rs:61-76      (f32, left-to-right, no FMA: build with -ffp-contract=off)
 *   find_max       src/mcts.rs:78-88      (Iterator::max_by -> LAST maximal element)
 *   backward       src/mcts.rs:90-98
 *   get_noise      src/mcts.rs:123-130    (Dirichlet(0.3); the RNG is ours: the reference uses thread_rng)
 *   select         src/mcts.rs:132-227
 *   mcts           src/mcts.rs:237-289
 *   step           src/mcts.rs:292-328
 *   predict        src/backends/torch.rs:89-146, post_process_distr src/chess.rs:879-903
 *   selfplay loop  src/main.rs:155-238
 */
#include "sc_oracle_mcts.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ shared exact helpers */
uint64_t orc_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
uint64_t orc_rng(uint64_t seed, uint64_t game, uint64_t ply, uint64_t purpose, uint64_t counter) {
    uint64_t h = orc_mix64(seed ^ (game * 0xD1B54A32D192ED03ULL));
    h = orc_mix64(h ^ (ply * 0x8CB92BA72F3D8DD7ULL));
    return orc_mix64(h ^ ((purpose << 48) | counter));
}
uint64_t orc_pos_hash(const orc_state* s) {
    const orc_pos* p = &s->cur;
    uint64_t h = 0x243F6A8885A308D3ULL;
    for (int sq = 0; sq < 64; sq++) {
        int8_t pc = p->board[sq];
        if (!pc) continue;
        uint64_t code = pc > 0 ? (uint64_t)pc : (uint64_t)(6 - pc);
        h = orc_mix64(h ^ ((code << 8) | (uint64_t)sq));
    }
    uint64_t tail = (uint64_t)p->turn | ((uint64_t)p->castling << 1) | ((uint64_t)(p->ep + 1) << 5) |
                    ((uint64_t)p->halfmove << 12);
    return orc_mix64(h ^ tail);
}

void orc_eval_synth(void* user, const orc_state* st, int n_legal, const orc_move* legal, const int* legal_idx,
                    float* priors, float* value) {
    (void)legal_idx;
    /* user: NULL, or a uint64 salt that makes a second, different deterministic "player" (match tests) */
    uint64_t h = orc_pos_hash(st) ^ (user ? *(const uint64_t*)user : 0);
    uint64_t sum = 0;
    uint32_t w[ORC_MAX_MOVES];
    for (int i = 0; i < n_legal; i++) {
        w[i] = 1u + (uint32_t)(orc_mix64(h ^ ((uint64_t)legal[i] * 0x9E3779B97F4A7C15ULL)) >> 40);
        sum += w[i];
    }
    float fs = (float)sum;
    for (int i = 0; i < n_legal; i++) priors[i] = (float)w[i] / fs;
    int64_t v = (int64_t)(orc_mix64(h ^ 0xABCDEFULL) >> 40) - 8388608;
    *value = (float)v / 8388608.0f;
}

/* Test evaluators that make exact PUCT ties (the hash evaluator's 24-bit priors never collide): find_max keeps the LAST
 * maximum (src/mcts.rs:78-88).  COARSE: 2-bit weights, values from {-0.5, 0, 0.5}; UNIFORM: equal priors, value 0. */
void orc_eval_synth_coarse(void* user, const orc_state* st, int n_legal, const orc_move* legal, const int* legal_idx,
                           float* priors, float* value) {
    (void)legal_idx;
    uint64_t h = orc_pos_hash(st) ^ (user ? *(const uint64_t*)user : 0);
    uint64_t sum = 0;
    uint32_t w[ORC_MAX_MOVES];
    for (int i = 0; i < n_legal; i++) {
        uint32_t full = 1u + (uint32_t)(orc_mix64(h ^ ((uint64_t)legal[i] * 0x9E3779B97F4A7C15ULL)) >> 40);
        w[i] = 1u + (full >> 22);
        sum += w[i];
    }
    float fs = (float)sum;
    for (int i = 0; i < n_legal; i++) priors[i] = (float)w[i] / fs;
    int64_t v = (int64_t)(orc_mix64(h ^ 0xABCDEFULL) >> 40) - 8388608;
    float fv = (float)v / 8388608.0f;
    *value = fv < -0.5f ? -0.5f : fv >= 0.5f ? 0.5f : 0.0f;
}
void orc_eval_synth_uniform(void* user, const orc_state* st, int n_legal, const orc_move* legal, const int* legal_idx,
                            float* priors, float* value) {
    (void)user;
    (void)st;
    (void)legal;
    (void)legal_idx;
    for (int i = 0; i < n_legal; i++) priors[i] = 1.0f / (float)n_legal;
    *value = 0.0f;
}

void orc_eval_net(void* user, const orc_state* st, int n_legal, const orc_move* legal, const int* legal_idx,
                  float* priors, float* value) {
    (void)legal;
    const orc_net* net = (const orc_net*)user;
    int8_t boards[8 * 8 * 112];
    int32_t meta[7];
    float logp[4672];
    orc_encode(st, boards, meta);
    orc_net_forward(net, boards, meta, logp, value, NULL);
    /* _get_move_distribution (torch.rs:148-175) + post_process_distr (chess.rs:891-901) */
    float sum = 0;
    for (int i = 0; i < n_legal; i++) {
        priors[i] = expf(logp[legal_idx[i]]);
        sum += priors[i];
    }
    sum += 1e-5f;
    for (int i = 0; i < n_legal; i++) priors[i] = priors[i] / sum;
}

/* ------------------------------------------------------------------ tree */
typedef struct {
    orc_move move;
    uint8_t color; /* side to move at this node (Step.1) */
    uint32_t depth;
    float q;
    int32_t n;
    float uct;
    float prior;
    int32_t parent, first_child, n_child;
} node_t;

struct orc_search {
    orc_state root;
    orc_state work;
    node_t* nodes;
    int n_nodes, cap;
    int32_t path[ORC_MAX_PLY];
    int path_len;
    int64_t n_evals;
    uint64_t rng;
};

static int new_node(orc_search* s) {
    if (s->n_nodes == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 1024;
        s->nodes = (node_t*)realloc(s->nodes, sizeof(node_t) * (size_t)s->cap);
    }
    return s->n_nodes++;
}
orc_search* orc_search_new(const orc_state* root, int root_depth) {
    orc_search* s = (orc_search*)calloc(1, sizeof *s);
    orc_state_copy(&s->root, root);
    int r = new_node(s);
    node_t* n = &s->nodes[r];
    memset(n, 0, sizeof *n);
    n->move = root->n ? root->moves[root->n - 1] : 0;
    n->color = root->cur.turn;
    n->depth = (uint32_t)root_depth;
    n->parent = -1;
    n->first_child = -1;
    s->rng = 0x1234567ULL;
    return s;
}
void orc_search_free(orc_search* s) {
    if (!s) return;
    free(s->nodes);
    free(s);
}
void orc_search_set_rng(orc_search* s, uint64_t seed) { s->rng = seed; }
int orc_search_num_nodes(const orc_search* s) { return s->n_nodes; }
int64_t orc_search_num_evals(const orc_search* s) { return s->n_evals; }
void orc_search_dump(const orc_search* s, int32_t* parent, uint16_t* move, int32_t* n, float* q, float* uct,
                     int32_t* first_child, int32_t* n_child) {
    for (int i = 0; i < s->n_nodes; i++) {
        const node_t* d = &s->nodes[i];
        if (parent) parent[i] = d->parent;
        if (move) move[i] = d->move;
        if (n) n[i] = d->n;
        if (q) q[i] = d->q;
        if (uct) uct[i] = d->uct;
        if (first_child) first_child[i] = d->first_child;
        if (n_child) n_child[i] = d->n_child;
    }
}
int orc_search_last_path(const orc_search* s, int32_t* path) {
    memcpy(path, s->path, sizeof(int32_t) * (size_t)s->path_len);
    return s->path_len;
}

/* ---- Dirichlet(0.3) from the oracle's own generator (distributional parity only) ---- */
static double u01(uint64_t* st) {
    *st = orc_mix64(*st);
    return ((double)(*st >> 11) + 0.5) / 9007199254740992.0;
}
static double gauss(uint64_t* st) {
    double a = u01(st), b = u01(st);
    return sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b);
}
static double gamma_sample(uint64_t* st, double alpha) {
    /* Marsaglia-Tsang, with the alpha<1 boost (what rand_distr::Gamma does for small shapes) */
    double boost = 1.0;
    if (alpha < 1.0) {
        boost = pow(u01(st), 1.0 / alpha);
        alpha += 1.0;
    }
    double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = gauss(st), v = 1.0 + c * x;
        if (v <= 0) continue;
        v = v * v * v;
        double u = u01(st);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return boost * d * v;
    }
}
static void dirichlet(uint64_t* st, int n, double* out) {
    double sum = 0;
    for (int i = 0; i < n; i++) {
        out[i] = gamma_sample(st, 0.3);
        sum += out[i];
    }
    for (int i = 0; i < n; i++) out[i] /= sum;
}

/* uct: src/mcts.rs:61-76 */
static float uct_fn(float sqrt_total, float prior, float move_q, int32_t move_n, int reverse_q, float cpuct) {
    float average_award = move_q / ((float)move_n + 1e-4f) * (reverse_q ? -1.0f : 1.0f);
    float exploration = (sqrt_total + 0.01f) / (1.0f + (float)move_n) * cpuct * prior;
    return average_award + exploration;
}

/* Game::predict (torch.rs:89-146) on the working state; returns number of steps */
static int predict(orc_search* s, orc_eval_fn eval, void* user, orc_move* legal, float* priors, float* value) {
    int n = orc_legal_moves(&s->work, legal);
    if (n == 0) {
        /* torch.rs:98-106: outcome().winner -> +1 white / -1 black / 0 */
        int term, winner;
        orc_outcome(&s->work, &term, &winner);
        *value = winner == 1 ? 1.0f : winner == 0 ? -1.0f : 0.0f;
        return 0;
    }
    int idx[ORC_MAX_MOVES];
    for (int i = 0; i < n; i++) idx[i] = orc_move_index(legal[i], s->work.cur.turn);
    eval(user, &s->work, n, legal, idx, priors, value);
    s->n_evals++;
    return n;
}

void orc_search_sim(orc_search* s, orc_eval_fn eval, void* user, float cpuct, float epsilon, int with_noise,
                    const double* noise_in, int faithful) {
    orc_state_copy(&s->work, &s->root); /* state.dup() mcts.rs:262 */
    s->path_len = 0;
    s->path[s->path_len++] = 0;
    orc_move legal[ORC_MAX_MOVES];
    float prior[ORC_MAX_MOVES];
    float outcome = 0;
    int n_steps = 0;
    for (;;) {
        int cur = s->path[s->path_len - 1];
        int have_pred = 0;
        if (faithful || s->nodes[cur].n_child == 0) {
            n_steps = predict(s, eval, user, legal, prior, &outcome);
            have_pred = 1;
        }
        node_t* nd = &s->nodes[cur];
        if (nd->n_child == 0 || (have_pred && n_steps == 0)) break;
        int nc = nd->n_child, fc = nd->first_child;
        if (!have_pred)
            for (int i = 0; i < nc; i++) prior[i] = s->nodes[fc + i].prior;
        int reverse_q = nd->color == ORC_BLACK; /* torch.rs:49-52 */
        int best;
        if (nc == 1) {
            best = 0;
        } else {
            int is_root = s->path_len == 1;
            float prior_rand[ORC_MAX_MOVES];
            if (!is_root || !with_noise) {
                memcpy(prior_rand, prior, sizeof(float) * (size_t)nc);
            } else {
                double noise_buf[ORC_MAX_MOVES];
                const double* noise = noise_in;
                if (!noise) {
                    dirichlet(&s->rng, nc, noise_buf);
                    noise = noise_buf;
                }
                for (int i = 0; i < nc; i++)
                    prior_rand[i] = prior[i] * (1.0f - epsilon) + (float)noise[i] * epsilon;
            }
            int32_t total = 0;
            for (int i = 0; i < nc; i++) total += s->nodes[fc + i].n;
            float sqrt_total = sqrtf((float)total);
            best = 0;
            float best_u = 0;
            for (int i = 0; i < nc; i++) {
                node_t* c = &s->nodes[fc + i];
                float u = uct_fn(sqrt_total, prior_rand[i], c->q, c->n, reverse_q, cpuct);
                c->uct = u;
                if (i == 0 || u >= best_u) { /* max_by keeps the LAST maximum */
                    best = i;
                    best_u = u;
                }
            }
        }
        orc_push(&s->work, s->nodes[fc + best].move); /* state.advance */
        s->path[s->path_len++] = fc + best;
    }
    /* expand (mcts.rs:267-283) */
    int leaf = s->path[s->path_len - 1];
    if (n_steps > 0) {
        int fc = s->n_nodes;
        for (int i = 0; i < n_steps; i++) {
            int id = new_node(s);
            node_t* c = &s->nodes[id];
            memset(c, 0, sizeof *c);
            c->move = legal[i];
            c->color = (uint8_t)!s->work.cur.turn;
            c->depth = s->nodes[leaf].depth + 1;
            c->prior = prior[i];
            c->parent = leaf;
            c->first_child = -1;
        }
        s->nodes[leaf].first_child = fc;
        s->nodes[leaf].n_child = n_steps;
    }
    /* backward (mcts.rs:90-98) */
    for (int i = 0; i < s->path_len; i++) {
        node_t* v = &s->nodes[s->path[i]];
        v->n += 1;
        v->q += outcome;
    }
}

int orc_choose_child(const int32_t* n_act, int n, float temp, float u01v) {
    if (n == 0) return -1;
    if (temp == 0.0f) {
        int32_t mx = n_act[0];
        for (int i = 1; i < n; i++)
            if (n_act[i] > mx) mx = n_act[i];
        for (int i = 0; i < n; i++)
            if (n_act[i] == mx) return i;
    }
    float power = 1.0f / temp;
    float cum[ORC_MAX_MOVES];
    float total = 0;
    for (int i = 0; i < n; i++) {
        float w = power == 1.0f ? (float)n_act[i] : powf((float)n_act[i], power);
        total += w;
        cum[i] = total;
    }
    float x = u01v * total;
    int idx = 0;
    for (int i = 0; i < n - 1; i++)
        if (cum[i] <= x) idx++;
    return idx;
}

/* ------------------------------------------------------------------ self-play (main.rs:155-238) */
orc_trace* orc_selfplay_game(const orc_selfplay_cfg* cfg, orc_eval_fn eval, void* user) {
    orc_trace* tr = (orc_trace*)calloc(1, sizeof *tr);
    int cap = 4096, used = 0;
    tr->child_move = (orc_move*)malloc(sizeof(orc_move) * (size_t)cap);
    tr->child_n = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    tr->child_q = (float*)malloc(sizeof(float) * (size_t)cap);
    tr->child_uct = (float*)malloc(sizeof(float) * (size_t)cap);
    orc_state* st = orc_state_new();
    int depth = 0;
    for (int i = 0; i < cfg->num_steps && i < ORC_MAX_PLY - 1; i++) {
        float temperature = i < cfg->temperature_switch ? 1.0f : cfg->temperature;
        orc_search* s = orc_search_new(st, depth);
        orc_search_set_rng(s, orc_rng(cfg->seed, cfg->game_id, (uint64_t)i, 2, 0));
        /* main.rs:175-180: (Some(v), None) => i32::min(300, (state.legal_moves().len() as f32 * v) as i32) */
        int rollout = cfg->rollout_num;
        if (cfg->rollout_factor > 0.0f) {
            orc_move lm[ORC_MAX_MOVES];
            int r = (int)((float)orc_legal_moves(st, lm) * cfg->rollout_factor);
            rollout = r < 300 ? r : 300;
        }
        for (int r = 0; r < rollout; r++)
            orc_search_sim(s, eval, user, cfg->cpuct, cfg->epsilon, cfg->with_noise, NULL, cfg->faithful);
        tr->n_sims += rollout;
        tr->n_evals += s->n_evals;
        node_t* root = &s->nodes[0];
        int nc = root->n_child;
        int32_t nact[ORC_MAX_MOVES];
        for (int c = 0; c < nc; c++) nact[c] = s->nodes[root->first_child + c].n;
        float u = (float)(orc_rng(cfg->seed, cfg->game_id, (uint64_t)i, 1, 0) >> 40) / 16777216.0f;
        int choice = orc_choose_child(nact, nc, temperature, u);
        if (choice < 0) { /* mcts::step -> None (main.rs:213-216) */
            tr->has_outcome = orc_outcome(st, &tr->termination, &tr->winner);
            orc_search_free(s);
            break;
        }
        if (used + nc > cap) {
            cap = (used + nc) * 2;
            tr->child_move = (orc_move*)realloc(tr->child_move, sizeof(orc_move) * (size_t)cap);
            tr->child_n = (int32_t*)realloc(tr->child_n, sizeof(int32_t) * (size_t)cap);
            tr->child_q = (float*)realloc(tr->child_q, sizeof(float) * (size_t)cap);
            tr->child_uct = (float*)realloc(tr->child_uct, sizeof(float) * (size_t)cap);
        }
        for (int c = 0; c < nc; c++) {
            node_t* ch = &s->nodes[root->first_child + c];
            tr->child_move[used + c] = ch->move;
            tr->child_n[used + c] = ch->n;
            tr->child_q[used + c] = ch->q;
            tr->child_uct[used + c] = ch->uct;
        }
        orc_move mv = s->nodes[root->first_child + choice].move;
        tr->moves[tr->n_steps] = mv;
        tr->q_root[tr->n_steps] = root->q;
        tr->child_off[tr->n_steps] = used;
        used += nc;
        tr->n_steps++;
        tr->child_off[tr->n_steps] = used;
        orc_push(st, mv);
        depth++;
        orc_search_free(s);
        if (i > cfg->outcome_gate) { /* main.rs:223-228 */
            tr->has_outcome = orc_outcome(st, &tr->termination, &tr->winner);
            if (tr->has_outcome) break;
        }
    }
    orc_state_free(st);
    return tr;
}
/* ------------------------------------------------------------------ match play (src/play.rs:241-343)
 * Two players share one tree cursor and alternate by ply (play_loop :318-343): White's evaluator searches on even
 * plies, Black's on odd ones; Dirichlet noise off (:250); temperature 1 while the root depth is below the switch
 * (:262-266); at temperature 0 a UNIFORMLY RANDOM child among those with the maximal visit count (:268-277; the
 * build-defined draw is floor(u01 * count) with the same uniform as the temperature sampling); the chosen child is
 * reset (step :300-301); outcome(claim_draw=True) is consulted after EVERY ply (:335), at most num_steps plies. */
orc_trace* orc_match_game(const orc_selfplay_cfg* cfg, orc_eval_fn eval_w, void* user_w, orc_eval_fn eval_b, void* user_b) {
    orc_trace* tr = (orc_trace*)calloc(1, sizeof *tr);
    int cap = 4096, used = 0;
    tr->child_move = (orc_move*)malloc(sizeof(orc_move) * (size_t)cap);
    tr->child_n = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    tr->child_q = (float*)malloc(sizeof(float) * (size_t)cap);
    tr->child_uct = (float*)malloc(sizeof(float) * (size_t)cap);
    orc_state* st = orc_state_new();
    for (int i = 0; i < cfg->num_steps && i < ORC_MAX_PLY - 1; i++) {
        orc_eval_fn eval = (i % 2 == 0) ? eval_w : eval_b;
        void* user = (i % 2 == 0) ? user_w : user_b;
        float temperature = i < cfg->temperature_switch ? 1.0f : cfg->temperature;
        orc_search* s = orc_search_new(st, i);
        for (int r = 0; r < cfg->rollout_num; r++) orc_search_sim(s, eval, user, cfg->cpuct, 0.15f, 0, NULL, cfg->faithful);
        tr->n_sims += cfg->rollout_num;
        tr->n_evals += s->n_evals;
        node_t* root = &s->nodes[0];
        int nc = root->n_child;
        if (nc == 0) { /* bestmove -> None: the reference unwraps and panics (:329); a finished game never gets here */
            orc_search_free(s);
            break;
        }
        int32_t nact[ORC_MAX_MOVES];
        for (int c = 0; c < nc; c++) nact[c] = s->nodes[root->first_child + c].n;
        float u = (float)(orc_rng(cfg->seed, cfg->game_id, (uint64_t)i, 1, 0) >> 40) / 16777216.0f;
        int choice;
        if (temperature == 0.0f) {
            int32_t mx = nact[0];
            int cnt = 0;
            for (int c = 1; c < nc; c++)
                if (nact[c] > mx) mx = nact[c];
            for (int c = 0; c < nc; c++) cnt += nact[c] == mx;
            int k = (int)(u * (float)cnt);
            if (k >= cnt) k = cnt - 1;
            choice = 0;
            for (int c = 0; c < nc; c++)
                if (nact[c] == mx && k-- == 0) {
                    choice = c;
                    break;
                }
        } else {
            choice = orc_choose_child(nact, nc, temperature, u);
        }
        if (used + nc > cap) {
            cap = (used + nc) * 2;
            tr->child_move = (orc_move*)realloc(tr->child_move, sizeof(orc_move) * (size_t)cap);
            tr->child_n = (int32_t*)realloc(tr->child_n, sizeof(int32_t) * (size_t)cap);
            tr->child_q = (float*)realloc(tr->child_q, sizeof(float) * (size_t)cap);
            tr->child_uct = (float*)realloc(tr->child_uct, sizeof(float) * (size_t)cap);
        }
        for (int c = 0; c < nc; c++) {
            node_t* ch = &s->nodes[root->first_child + c];
            tr->child_move[used + c] = ch->move;
            tr->child_n[used + c] = ch->n;
            tr->child_q[used + c] = ch->q;
            tr->child_uct[used + c] = ch->uct;
        }
        orc_move mv = s->nodes[root->first_child + choice].move;
        tr->moves[tr->n_steps] = mv;
        tr->q_root[tr->n_steps] = root->q;
        tr->child_off[tr->n_steps] = used;
        used += nc;
        tr->n_steps++;
        tr->child_off[tr->n_steps] = used;
        orc_push(st, mv);
        orc_search_free(s);
        tr->has_outcome = orc_outcome(st, &tr->termination, &tr->winner);
        if (tr->has_outcome) break;
    }
    orc_state_free(st);
    return tr;
}
void orc_trace_free(orc_trace* t) {
    if (!t) return;
    free(t->child_move);
    free(t->child_n);
    free(t->child_q);
    free(t->child_uct);
    free(t);
}
