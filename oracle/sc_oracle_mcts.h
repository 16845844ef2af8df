/* sc_oracle_mcts.h -- search / predict / self-play part of the CPU ORACLE (test infrastructure only). */
#ifndef SC_ORACLE_MCTS_H
#define SC_ORACLE_MCTS_H
#include "sc_oracle.h"
#include "sc_oracle_nn.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Game::predict tail (src/backends/torch.rs:108-146): given the state and its legal moves
 * (python-chess order) and their action indices, return renormalised priors and the value from
 * White's point of view. */
typedef void (*orc_eval_fn)(void* user, const orc_state* st, int n_legal, const orc_move* legal,
                            const int* legal_idx, float* priors, float* value);

/* built-in evaluators */
void orc_eval_synth(void* user, const orc_state*, int, const orc_move*, const int*, float*, float*); /* integer-hash priors/values, bit-reproducible on the GPU */
void orc_eval_synth_coarse(void* user, const orc_state*, int, const orc_move*, const int*, float*, float*);  /* 2-bit priors, values {-0.5,0,0.5}: exact ties */
void orc_eval_synth_uniform(void* user, const orc_state*, int, const orc_move*, const int*, float*, float*); /* uniform priors, value 0: every unvisited sibling ties */
void orc_eval_net(void* user /* orc_net* */, const orc_state*, int, const orc_move*, const int*, float*, float*);

/* shared, exactly specified helpers (the engine implements the same functions) */
uint64_t orc_mix64(uint64_t);
uint64_t orc_pos_hash(const orc_state*);
uint64_t orc_rng(uint64_t seed, uint64_t game, uint64_t ply, uint64_t purpose, uint64_t counter);

typedef struct orc_search orc_search;
orc_search* orc_search_new(const orc_state* root, int root_depth);
void orc_search_free(orc_search*);
/* one iteration of src/mcts.rs:261-288 (dup, select, expand, backward).
 * noise: NULL -> draw Dirichlet(0.3) from the oracle's own RNG when with_noise; else n_root_children doubles.
 * faithful != 0: call the evaluator at every node of the descent (mcts.rs:152); 0: cached priors. */
void orc_search_sim(orc_search*, orc_eval_fn eval, void* user, float cpuct, float epsilon, int with_noise,
                    const double* noise, int faithful);
int orc_search_num_nodes(const orc_search*);
int64_t orc_search_num_evals(const orc_search*);
/* node arrays in allocation order (root = 0; children of an expanded node are contiguous) */
void orc_search_dump(const orc_search*, int32_t* parent, uint16_t* move, int32_t* n, float* q, float* uct,
                     int32_t* first_child, int32_t* n_child);
int orc_search_last_path(const orc_search*, int32_t* path); /* nodes visited by the last simulation */
void orc_search_set_rng(orc_search*, uint64_t seed);

/* mcts::step (src/mcts.rs:292-328): temp==0 -> first max-N child; else sample ~ N^(1/temp) with
 * u01 (the build-defined uniform in [0,1) as float with 24 random bits). Returns child index or -1. */
int orc_choose_child(const int32_t* n_act, int n, float temp, float u01);

typedef struct {
    int n_steps;
    orc_move moves[ORC_MAX_PLY];
    float q_root[ORC_MAX_PLY];
    int child_off[ORC_MAX_PLY + 1];
    orc_move* child_move; /* malloc'ed, child_off[n_steps] entries */
    int32_t* child_n;
    float* child_q;
    float* child_uct;
    int has_outcome, termination, winner;
    int64_t n_sims, n_evals;
} orc_trace;

typedef struct {
    int rollout_num;        /* --rollout-num */
    int num_steps;          /* -n */
    float cpuct;            /* --cpuct */
    float temperature;      /* --temperature */
    int temperature_switch; /* --temperature-switch */
    float epsilon;          /* --epsilon */
    int with_noise;         /* selfplay: 1 (src/main.rs:195) */
    int faithful;           /* re-evaluate at every node like the reference */
    uint64_t seed;
    uint64_t game_id;
    int outcome_gate;       /* src/main.rs:223: outcome is only looked at when i > 100 */
    float rollout_factor;   /* > 0: --rollout-factor, per-ply budget min(300, n_legal * factor) (src/main.rs:175-176) */
} orc_selfplay_cfg;

/* src/main.rs:155-238 */
orc_trace* orc_selfplay_game(const orc_selfplay_cfg*, orc_eval_fn eval, void* user);
/* src/play.rs:241-343 (two players, shared cursor, no noise, random tie-break, outcome after every ply); with_noise,
 * epsilon and outcome_gate of the cfg are ignored */
orc_trace* orc_match_game(const orc_selfplay_cfg*, orc_eval_fn eval_white, void* user_white, orc_eval_fn eval_black, void* user_black);
void orc_trace_free(orc_trace*);

#ifdef __cplusplus
}
#endif
#endif
