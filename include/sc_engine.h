/*
 * sc_engine.h -- C ABI of libsc_engine.so, the MI355X-native drop-in for the MCTS + NN rollout
 * hot path of pierric/smart-chess-rust.
 *
 * Plain pointers and sizes only (no torch / C++ types).  Every entry point returns 0 on success
 * and a negative code on failure; sc_last_error() gives the thread-local message.  Nothing aborts
 * across the ABI (the reference unwrap()s/panics instead: src/backends/torch.rs:27-31,100,111).
 * A handle is bound to one GPU and must be driven by one host thread at a time; distinct handles
 * (one per GPU) are independent -- this is how games shard over the 8 GPUs of a node.
 *
 * Two levels, as laid out in SURVEY.md section 8(b):
 *
 *  L-predict  -- the reference's `trait Game<S>::predict` contract (src/game.rs:3-15) as
 *                implemented by src/backends/torch.rs:89-146 / src/backends/onnx.rs:14-56,
 *                batched.  The reference-side binding is a `impl Game<BoardState> for ChessHip`
 *                (INTEGRATION.md).
 *  L-search   -- the whole per-game loop of src/main.rs:155-238 (mcts::mcts src/mcts.rs:237-289,
 *                mcts::step :292-328, Trace src/trace.rs:5-42) for many concurrent games on one GPU.
 *
 * Layout conventions (identical to the reference's post-_encode tensors):
 *   boards : int8  [n][8][8][112]  (rank, file, plane)      src/chess.rs:828-842, :845-877
 *   meta   : int32 [n][7]                                    src/chess.rs:652-662
 *   moves  : uint16 = from | to<<6 | promo<<12, squares a1=0..h8=63, promo in python-chess piece
 *            types (0 none, 2 N, 3 B, 4 R, 5 Q); castling is the king's two-square move (e1g1)
 *   action index: rank*584 + file*73 + type after rotating Black's moves (src/chess.rs:504-551)
 */
#ifndef SC_ENGINE_H
#define SC_ENGINE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_MAX_MOVES 224      /* row stride of per-position move tables (218 is the chess maximum) */
#define SC_POLICY_SIZE 4672   /* 8*8*73 */
#define SC_BOARD_BYTES 7168   /* 8*8*112 */

typedef struct sc_engine sc_engine;
typedef struct sc_selfplay sc_selfplay;

const char* sc_last_error(void);
/* Return codes (every entry point: 0 = ok, or a small positive "not yet / not there" code where its comment says so):
 *   -1 bad argument or request, -2 HIP runtime error, -3 no GPU (there is no CPU fallback),
 *   SC_ERR_HANDOFF: the self-play handle is POISONED -- an internal hand-off of one of its step launches timed out
 *   (sc_selfplay_stats.error_flags & 48), so values were computed from stale data and its trees / traces are invalid.  Once the
 *   host has seen that (any synchronising call), sc_selfplay_enqueue_sims / _enqueue_interleaved / _run / _synchronize / _poll /
 *   _get_trace / _write_trace_json refuse with this code; sc_selfplay_get_stats still answers (the flags).  Destroy the handle;
 *   handles created afterwards on that device use the two-launch step, which has no hand-off between workgroups. */
#define SC_ERR_HANDOFF (-5)
int sc_device_count(void);
/* Launch-path state of the HIP runtime in this process, bit mask.  The engine's three launches per simulation step are
 * ~6 % faster with kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1), which the HIP runtime reads ONCE, when
 * it initialises:
 *   bit 0  HIP_FORCE_DEV_KERNARG=1 is in the process environment
 *   bit 1  ... and it was put there by this library's load hook (the host had not set it): it is effective only if the
 *          host had not initialised HIP before loading libsc_engine.so -- which the library cannot observe.  Hosts that
 *          load torch/tch first should export the variable themselves; then bit 1 is clear.
 *   bit 2  the load hook was disabled by SC_ENGINE_KEEP_ENV=1 (the library never touches the environment)
 * sc_engine_create leaves a note in sc_last_warning() when bit 0 is clear or bit 1 is set. */
int sc_runtime_flags(void);
const char* sc_last_warning(void);

/* ------------------------------------------------------------------ network (L-predict) */
enum { SC_PREC_BF16 = 0, SC_PREC_FP8 = 1 };
typedef struct {
    int32_t n_res_blocks;  /* py/module.py:110 (reference default 19) */
    int32_t channels;      /* trunk width: 256 in the reference (py/module.py:120-133); 128 = BASELINE cfg2 variant */
    uint64_t seed;         /* used when weights_path == NULL: build-owned deterministic init (tools/scw.py) */
    int32_t precision;     /* SC_PREC_BF16: the reference's exported precision (py/export.py:47-65, bf16 autocast).
                              SC_PREC_FP8 (BASELINE configs[4]): the convs (stem, blocks, head convs) run on the CDNA4 fp8
                              matrix cores -- OCP e4m3 operands, per-output-channel power-of-two weight scales, fp32
                              accumulate; LayerNorm / residual stream / softmax fp32 and the SE + value FC layers bf16 as
                              before.  Stated tolerance vs the fp32 reference vectors: prior total
                              variation < 0.05, |value| error < 0.05. */
    int32_t reserved;
} sc_net_config;

/* Replaces backend construction in src/main.rs:83-128 (ChessTS / ChessEP / ChessOnnx).
 * weights_path: NULL, or a blob written by tools/scw.py / tools/ckpt_to_scw.py from a reference state_dict: SCW1 (fp32
 * tensors; quantised at load as cfg->precision asks) or SCW2 (the fp8 export: e4m3 conv weights + their channel scales;
 * the blob's precision wins). */
int sc_engine_create(const sc_net_config* cfg, const char* weights_path, int device_id, sc_engine** out);
void sc_engine_destroy(sc_engine*);
int sc_engine_max_batch(const sc_engine*);
int sc_engine_precision(const sc_engine*);   /* SC_PREC_* the engine runs in */

/* ChessModule.forward on a batch (src/backends/torch.rs:115-125, py/module.py:135-154):
 * host buffers in, logp [n][4672] fp32 (log-softmax, channel-major flatten) and value [n] out
 * (value from White's point of view).  logp may be NULL. */
int sc_forward_batch(sc_engine*, int n, const int8_t* boards, const int32_t* meta, float* logp, float* value);

/* The post-_encode tail of Game::predict (src/backends/torch.rs:108-146): forward, gather the
 * legal action logits, exp, renormalise by (sum + 1e-5) (src/chess.rs:891-901).
 * legal_idx / priors are CSR: position i owns [legal_off[i], legal_off[i+1]). */
int sc_predict_batch(sc_engine*, int n, const int8_t* boards, const int32_t* meta, const uint16_t* legal_idx,
                     const uint32_t* legal_off, float* priors, float* value);

/* Game::predict with argmax = true (the reference passes it from chess_play_inference, src/lib.rs:333-337): the same, then
 * post_process_distr's argmax branch (src/chess.rs:880-889) -- priors become a one-hot vector at the LAST maximum of each
 * position (Iterator::max_by). */
int sc_predict_batch_argmax(sc_engine*, int n, const int8_t* boards, const int32_t* meta, const uint16_t* legal_idx,
                            const uint32_t* legal_off, float* priors, float* value);
int sc_engine_synchronize(sc_engine*);
/* test aid: fp32 residual stream [n][64][channels] after `stage` (0 = conv_block, b = res block b, 1000 = trunk output) */
int sc_forward_debug(sc_engine*, int n, const int8_t* boards, const int32_t* meta, int stage, float* out);

/* Rules + encoder on the GPU: replaces python-chess (src/chess.rs:665-803) and _encode
 * (src/chess.rs:845-877) for positions given as move lists from the start position.
 * For position i (moves[move_off[i]..move_off[i+1])):
 *   boards[i], meta[i]                      the NN input
 *   legal_moves[i][..], legal_idx[i][..]    legal moves in python-chess generation order and their
 *                                           action indices (row stride SC_MAX_MOVES), n_legal[i]
 *   outcome[i][0] = termination (src/chess.rs:88-99 numbering, 0 = none; outcome(claim_draw=True)),
 *   outcome[i][1] = winner (1 white, 0 black, -1 none), outcome[i][2] = is_check, outcome[i][3] = 0 ok / <0 illegal move at that index-1
 * Any output pointer may be NULL. */
int sc_encode_positions(sc_engine* engine_or_null, int device_id, int n, const uint16_t* moves, const uint32_t* move_off,
                        int8_t* boards, int32_t* meta, uint16_t* legal_moves, uint16_t* legal_idx, int32_t* n_legal,
                        int32_t* outcome);

/* Trace -> training tensors on the GPU: replaces libsmartchess.chess_encode_steps
 * (reference src/lib.rs:46-128; consumer py/dataset.py:47-87) for a batch of recorded games.
 *   moves / move_off     the played moves of game g: moves[move_off[g] .. move_off[g+1]); ply p (global index, game after
 *                        game) is the position BEFORE moves[p]
 *   child_mv / child_n / child_off   the searched children of ply p and their visit counts: [child_off[p], child_off[p+1])
 *                        (a trace's steps[i][2] = [[uci, N, Q, uct], ...]; at most SC_MAX_MOVES per ply)
 *   apply_mirror         the reference's colour-mirror augmentation (changes meta only; py/dataset.py negates the outcome)
 * Outputs, host pointers, any of the first five may be NULL (P = move_off[n_games] plies):
 *   boards int8[P][8][8][112], meta int32[P][7], dist float[P][4672] = N_i / (sum N + 1e-5) at the action index of
 *   the real mover, legal_idx uint16[P][SC_MAX_MOVES] + n_legal int32[P] (the reference's `move_indices`),
 *   status int32[n_games]: 0 ok; 1000+i: the children of ply i are not exactly the legal moves ("inconsistent moves",
 *   the reference panics, lib.rs:64-76); -(i+1): the move played at ply i is not legal (lib.rs:78-80).  Outputs of a
 *   game at and after its failing ply are unspecified. */
int sc_encode_steps(sc_engine* engine_or_null, int device_id, int n_games, const uint16_t* moves, const uint32_t* move_off,
                    const uint16_t* child_mv, const uint32_t* child_n, const uint32_t* child_off, int apply_mirror,
                    int8_t* boards, int32_t* meta, float* dist, uint16_t* legal_idx, int32_t* n_legal, int32_t* status);

/* measurement aid (bench.py also_encode_steps): the last sc_encode_steps call of the calling thread -- HIP-event time of its
 * kernels (game walk, keys, repetition flags, planes + moves, dist: all chunks) and wall time of the whole call including the PCIe copies */
int sc_encode_steps_last_timing(float* kernels_ms, float* total_ms);

/* ------------------------------------------------------------------ self-play (L-search) */
/* SYNTH: integer-hash evaluator for exact search-parity tests; SYNTH_COARSE: the same with 2-bit priors and values from
 * {-0.5, 0, 0.5} (exact PUCT ties between some siblings); SYNTH_UNIFORM: uniform priors, value 0 (every unvisited sibling
 * ties: find_max's last-maximum rule, src/mcts.rs:78-88, decides every descent) */
enum { SC_EVAL_NET = 0, SC_EVAL_SYNTH = 1, SC_EVAL_SYNTH_COARSE = 2, SC_EVAL_SYNTH_UNIFORM = 3 };

typedef struct {
    int32_t n_slots;            /* concurrent games on this GPU (BASELINE cfg2: 256) */
    int32_t n_games;            /* total games to play on this handle (slots are recycled) */
    int32_t rollout_num;        /* --rollout-num      src/main.rs:32-33,175-180 */
    int32_t num_steps;          /* -n/--num-steps     src/main.rs:35-36 */
    float cpuct;                /* --cpuct            src/main.rs:50-51 */
    float temperature;          /* --temperature      src/main.rs:47-48 */
    int32_t temperature_switch; /* --temperature-switch src/main.rs:53-54 */
    float epsilon;              /* --epsilon          src/main.rs:56-57 */
    int32_t with_noise;         /* 1 in selfplay (src/main.rs:195), 0 for NNPlayer::bestmove (src/play.rs:250) */
    int32_t outcome_gate;       /* outcome() is consulted only when ply index > gate (src/main.rs:223: 100) */
    int32_t evaluator;          /* SC_EVAL_NET / SC_EVAL_SYNTH */
    int32_t external_noise;     /* tests: root noise is taken from sc_selfplay_set_noise instead of the device RNG */
    uint64_t seed;
    uint64_t first_game_id;     /* global id of this handle's first game (sharding across GPUs/ranks) */
    int32_t trace_capacity;     /* traces kept on the device: 0 = n_games (every trace retrievable); >0 = ring of that
                                   many games (>= 2*n_slots): game k uses row k % capacity.  A row is never taken while
                                   its previous game is still being played (the new game waits); a FINISHED trace is
                                   overwritten (throughput runs) unless trace_hold is set */
    int32_t own_stream;         /* 1: this handle launches on its own HIP stream, so several handles (groups of games)
                                   of one engine overlap on the GPU: one group's tree work hides under another's network */
    int32_t tie_random;         /* temperature 0: 0 = first most-visited child (mcts::step, src/mcts.rs:298-306);
                                   1 = uniformly random among the most visited (NNPlayer::bestmove, src/play.rs:268-277) */
    int32_t trace_hold;         /* 1: a finished trace stays in the ring until sc_selfplay_poll has handed it to the host and
                                   the host has polled again; new games wait for a free row (streaming drain: the reference
                                   writes each trace file when its game ends, src/main.rs:235-238) */
    float rollout_factor;       /* > 0: -r/--rollout-factor (src/main.rs:29-30,175-176): every ply searches
                                   min(300, (n_legal * factor) as i32) simulations, n_legal = legal moves of the ply's root;
                                   rollout_num must then be 300 (it sizes the node pools) */
} sc_selfplay_config;

int sc_selfplay_create(sc_engine* engine_or_null, int device_id, const sc_selfplay_config* cfg, sc_selfplay** out);
void sc_selfplay_destroy(sc_selfplay*);

/* Enqueue `n` simulation steps (each = one iteration of src/mcts.rs:261-288 for every active
 * game, including the per-ply move choice of mcts::step when a game's rollout count is reached). */
int sc_selfplay_enqueue_sims(sc_selfplay*, int n);
int sc_selfplay_synchronize(sc_selfplay*);
/* Enqueue n simulation steps on several handles of one GPU, interleaved step by step (handles created with
 * own_stream = 1 overlap on the device). */
int sc_selfplay_enqueue_interleaved(sc_selfplay** handles, int n_handles, int n);
/* Run until every game has finished (or max_sim_steps > 0 is reached). */
int sc_selfplay_run(sc_selfplay*, int64_t max_sim_steps);

typedef struct {
    int64_t sims_done;       /* simulations completed, all games */
    int64_t nn_evals;        /* leaf evaluations that needed the network */
    int32_t games_finished;
    int32_t games_active;
    int32_t error_flags;     /* bit mask: 1 non-finite PUCT value (reference panics: src/mcts.rs:202-214), 2 node pool overflow,
                                4 move without action index, 8 descent deeper than the path buffer, 16 / 32 internal hand-off timeout (search helper / value-head tiles) */
    int32_t plies_done;      /* total plies played over all games */
} sc_selfplay_stats;
int sc_selfplay_get_stats(sc_selfplay*, sc_selfplay_stats* out);
/* HIP-event timing on the stream the kernels are launched on.
 * enable_timing(stride > 0): every stride-th simulation step runs as three separate launches with the network TOWER launch
 *   bracketed by an event pair (the tower-only figure).
 * enable_timing(stride < 0): every |stride|-th simulation step is bracketed AS A WHOLE by an event pair, in whatever launch form
 *   the handle uses (one k_step launch in the production form): the duration of the dominant kernel as production runs it.
 * enable_timing(0): off.
 * timing(): ms_total = first enqueue -> last enqueue span; ms_nn = sum over the nn_launches sampled brackets (at most the last
 * 4096). */
int sc_selfplay_enable_timing(sc_selfplay*, int stride);
/* Match play (the `play` binary's loop, src/play.rs:318-343; batched: every slot is one game of the same pairing).
 * After this call the handle alternates players by ply: even plies are searched with `white`, odd plies with `black`
 * (engines for SC_EVAL_NET; for SC_EVAL_SYNTH the two salts select two deterministic synthetic players).
 * Needs n_games == n_slots (all games advance in lockstep, no slot recycling) and must precede the first enqueue.
 * Reference settings: with_noise = 0, outcome_gate = -1 (outcome after every ply), num_steps = 200, tie_random = 1. */
int sc_selfplay_set_players(sc_selfplay*, sc_engine* white, sc_engine* black, uint64_t synth_salt_white, uint64_t synth_salt_black);
int sc_selfplay_timing(sc_selfplay*, int reset, float* ms_total, float* ms_nn, int64_t* nn_launches);
/* Kernel launches per simulation step this handle uses with SC_EVAL_NET (steps bracketed for sc_selfplay_timing always use 3):
 * 1 = the fused step kernel with value_head.ffn.0 inside (whole 64-slot blocks, every workgroup resident, and no other
 *     stream of this process running that form on the device: its workgroups wait for each other inside the launch),
 * 2 = the fused step kernel + the value FC launch, 3 = search, network tower and value FC as separate launches.
 * All three play bit-identical games.  0 for handles without a network. */
int sc_selfplay_launches_per_step(const sc_selfplay*);

/* Trace of a finished game = the reference's Trace<M,O> (src/trace.rs:5-9) in SoA form.
 * Call with NULL arrays to query sizes first.  child_off has n_steps+1 entries. */
typedef struct {
    int32_t n_steps;
    int32_t n_children_total;
    int32_t has_outcome;   /* 0: outcome null (src/trace.rs:7), 1: set */
    int32_t termination;   /* src/chess.rs:88-99 */
    int32_t winner;        /* 1 white, 0 black, -1 none */
    uint64_t game_id;
} sc_trace_info;
/* Returns 0 ok, 1 the game has not finished yet, 2 its trace is no longer on the device (ring row overwritten by a
 * later game, or released by sc_selfplay_poll), < 0 error. */
int sc_selfplay_get_trace(sc_selfplay*, int game /*0..n_games-1*/, sc_trace_info* info, uint16_t* step_move,
                          float* step_q, int32_t* child_off, uint16_t* child_move, int32_t* child_n, float* child_q,
                          float* child_uct);
/* Streaming drain: completes the enqueued work, then returns the number of games (written to
 * finished_games[0..cap), handle-local indices usable with sc_selfplay_get_trace / _write_trace_json) that have finished
 * since they were last reported.  With trace_hold the traces reported by the PREVIOUS call are released first (their ring
 * rows become free for new games), so the host reads each batch between two polls.  < 0: error. */
int sc_selfplay_poll(sc_selfplay*, int32_t* finished_games, int cap);
/* Writes the reference's trace JSON (src/trace.rs:23-32; serde_json pretty, keys "outcome","steps"). */
int sc_selfplay_write_trace_json(sc_selfplay*, int game, const char* path);

/* tests / NNPlayer::bestmove (src/play.rs:241-288) support: current search tree of a slot in
 * allocation order (root = 0; children of a node contiguous).  Arrays may be NULL; returns n_nodes. */
int sc_selfplay_get_tree(sc_selfplay*, int slot, int cap, int32_t* n, float* q, float* uct, float* prior, uint16_t* move,
                         int32_t* first_child, int32_t* n_child);
int sc_selfplay_get_slot(sc_selfplay*, int slot, int32_t* ply, int32_t* sim, int32_t* status, uint64_t* game_id,
                         int32_t* last_path /*cap 1024*/, int32_t* last_path_len);
/* replace the root noise used by the NEXT simulation of `slot` (external_noise mode); noise[n] */
int sc_selfplay_set_noise(sc_selfplay*, int slot, const float* noise, int n);
int sc_selfplay_get_noise(sc_selfplay*, int slot, float* noise, int cap);
/* start slot from a given move list instead of the initial position (sc_search / chess_play_new, src/lib.rs:161-232) */
int sc_selfplay_set_position(sc_selfplay*, int slot, const uint16_t* moves, int n_moves);
/* per-call search options of chess_play_mcts(state, rollout, cpuct, noise) (src/lib.rs:233-247): applies to the
 * simulations enqueued after the call */
int sc_selfplay_set_search(sc_selfplay*, float cpuct, float epsilon, int with_noise);

/* One search as a single call: NNPlayer::bestmove's mcts::mcts (src/play.rs:241-252) / chess_play_mcts (src/lib.rs:233-247).
 * Runs `rollout` simulations from the position reached by `moves` (fresh tree, epsilon 0.15) and returns the number of root
 * children (< 0: error); child_move / child_n / child_q / child_prior receive up to `cap` of them in python-chess move
 * order (any may be NULL), *root_q the root's value sum.  The engine keeps ONE one-slot handle for these calls and reuses it
 * (every call restarts from a one-node tree with its own options and seed: the result does not depend on earlier calls), so
 * calls on one engine must not overlap; to keep the tree between moves use a handle of your own (sc_selfplay_set_position +
 * sc_selfplay_enqueue_sims + sc_selfplay_get_tree). */
int sc_search(sc_engine*, const uint16_t* moves, int n_moves, int rollout, float cpuct, int with_noise, uint64_t seed, int cap,
              uint16_t* child_move, int32_t* child_n, float* child_q, float* child_prior, float* root_q);

/* test aid: find_max (src/mcts.rs:78-88, Iterator::max_by: the LAST maximum wins) as the descent computes it, on n <= 256
 * caller-provided finite values: out[0] = the one-round form used for nodes with <= 64 children (-2 if n > 64),
 * out[1] = the four-round (value, index) form used for wider nodes. */
int sc_debug_find_max(int device_id, const float* values, int n, int32_t* out2);
/* test aid: makes the NEXT one-launch steps of the handle wait for arrivals that never come (the in-launch hand-off's target is
 * raised by `missing` arrivals per block), to show on hardware that the wait is bounded: every workgroup gives up after ~0.2 s,
 * the launch ends and error_flags carries bit 32.  The handle's results are invalid afterwards and it refuses further work
 * (SC_ERR_HANDOFF).  No effect (returns 1) on a handle that does not use the one-launch form. */
int sc_selfplay_debug_break_handoff(sc_selfplay*, int missing);
/* test aid: forget that an in-launch hand-off has failed on the device (after such a failure new handles get the two-launch
 * step for the rest of the process; tests that provoke the failure restore the default with this).  0 = a record was cleared. */
int sc_debug_clear_handoff_failure(int device_id);
/* developer aid: stamps of the last launch, out[n_slots][32]: 0..7 the search's cycle stamps (tools/dbg_cycles.py); 24..27 the
 * one-launch step's phases on the 100 MHz wall clock (kernel entry, leaf selected, network done, value-FC tile done: bench.py's
 * per-phase split); the rest written by experiment builds only (tools/dbg_tail.py, tools/dbg_expand.py) */
int sc_selfplay_debug_cycles(sc_selfplay*, int enable, unsigned long long* out);

/* utility: trace-file JSON writer on caller-provided arrays (no GPU needed) */
int sc_trace_write_json(const char* path, const sc_trace_info* info, const uint16_t* step_move, const float* step_q,
                        const int32_t* child_off, const uint16_t* child_move, const int32_t* child_n,
                        const float* child_q, const float* child_uct);
/* utility: UCI text of a move (src/chess.rs:513-519); returns strlen */
int sc_move_uci(uint16_t move, char* buf8);
/* libsmartchess.chess_encode_move(turn, move) (reference src/lib.rs:37-44; src/chess.rs:544-550, queenmoves.rs,
 * knightmoves.rs, underpromotions.rs): action index in [0, 4672) of `move` for the side to move, -1 if it has none.
 * Host function, needs no GPU. */
int sc_move_index(uint16_t move, int white_to_move);

#ifdef __cplusplus
}
#endif
#endif
