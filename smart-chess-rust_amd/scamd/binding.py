"""ctypes binding of include/sc_engine.h (libsc_engine.so)."""
import ctypes as C
import os

# see INTEGRATION.md: kernel arguments in device memory (must precede HIP runtime initialisation)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SC_ENGINE_LIB: developer override (experiment builds of the same library); there is still no CPU fallback
_LIB_PATH = os.environ.get("SC_ENGINE_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libsc_engine.so")

MAX_MOVES = 224
TERMINATION = {0: None, 1: "Checkmate", 2: "Stalemate", 3: "InsufficientMaterial", 4: "SeventyfiveMoves",
               5: "FivefoldRepetition", 6: "FiftyMoves", 7: "ThreefoldRepetition"}
_WINNER = {1: "White", 0: "Black", -1: None}
EVALUATORS = {"net": 0, "synth": 1, "synth_coarse": 2, "synth_uniform": 3}   # SC_EVAL_* (include/sc_engine.h)


class EngineError(RuntimeError):
    code = None


ERR_HANDOFF = -5   # SC_ERR_HANDOFF: the self-play handle is poisoned (include/sc_engine.h)


class NetConfig(C.Structure):
    _fields_ = [("n_res_blocks", C.c_int32), ("channels", C.c_int32), ("seed", C.c_uint64), ("precision", C.c_int32),
                ("reserved", C.c_int32)]


class SelfplayConfig(C.Structure):
    _fields_ = [("n_slots", C.c_int32), ("n_games", C.c_int32), ("rollout_num", C.c_int32), ("num_steps", C.c_int32),
                ("cpuct", C.c_float), ("temperature", C.c_float), ("temperature_switch", C.c_int32),
                ("epsilon", C.c_float), ("with_noise", C.c_int32), ("outcome_gate", C.c_int32),
                ("evaluator", C.c_int32), ("external_noise", C.c_int32), ("seed", C.c_uint64),
                ("first_game_id", C.c_uint64), ("trace_capacity", C.c_int32), ("own_stream", C.c_int32),
                ("tie_random", C.c_int32), ("trace_hold", C.c_int32), ("rollout_factor", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("sims_done", C.c_int64), ("nn_evals", C.c_int64), ("games_finished", C.c_int32),
                ("games_active", C.c_int32), ("error_flags", C.c_int32), ("plies_done", C.c_int32)]


class TraceInfo(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("n_children_total", C.c_int32), ("has_outcome", C.c_int32),
                ("termination", C.c_int32), ("winner", C.c_int32), ("game_id", C.c_uint64)]


# every symbol include/sc_engine.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _u16p, _f = C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_float
ABI = {
    "sc_last_error": (C.c_char_p, []),
    "sc_device_count": (_i, []),
    "sc_runtime_flags": (_i, []),
    "sc_last_warning": (C.c_char_p, []),
    "sc_selfplay_poll": (_i, [_vp, _vp, _i]),
    "sc_debug_find_max": (_i, [_i, _vp, _i, _vp]),
    "sc_selfplay_debug_break_handoff": (_i, [_vp, _i]),
    "sc_debug_clear_handoff_failure": (_i, [_i]),
    "sc_selfplay_debug_cycles": (_i, [_vp, _i, _vp]),
    "sc_engine_create": (_i, [C.POINTER(NetConfig), C.c_char_p, _i, C.POINTER(_vp)]),
    "sc_engine_destroy": (None, [_vp]),
    "sc_engine_max_batch": (_i, [_vp]),
    "sc_engine_precision": (_i, [_vp]),
    "sc_forward_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "sc_predict_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_predict_batch_argmax": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_engine_synchronize": (_i, [_vp]),
    "sc_forward_debug": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "sc_encode_positions": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_selfplay_create": (_i, [_vp, _i, C.POINTER(SelfplayConfig), C.POINTER(_vp)]),
    "sc_selfplay_destroy": (None, [_vp]),
    "sc_selfplay_enqueue_sims": (_i, [_vp, _i]),
    "sc_selfplay_synchronize": (_i, [_vp]),
    "sc_selfplay_enqueue_interleaved": (_i, [_vp, _i, _i]),
    "sc_selfplay_run": (_i, [_vp, _i64]),
    "sc_selfplay_get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "sc_search": (_i, [_vp, _vp, _i, _i, _f, _i, C.c_uint64, _i, _vp, _vp, _vp, _vp, _vp]),
    "sc_selfplay_set_search": (_i, [_vp, _f, _f, _i]),
    "sc_selfplay_set_players": (_i, [_vp, _vp, _vp, C.c_uint64, C.c_uint64]),
    "sc_selfplay_enable_timing": (_i, [_vp, _i]),
    "sc_selfplay_timing": (_i, [_vp, _i, C.POINTER(_f), C.POINTER(_f), C.POINTER(_i64)]),
    "sc_selfplay_launches_per_step": (_i, [_vp]),
    "sc_selfplay_get_trace": (_i, [_vp, _i, C.POINTER(TraceInfo), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_selfplay_write_trace_json": (_i, [_vp, _i, C.c_char_p]),
    "sc_selfplay_get_tree": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_selfplay_get_slot": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_selfplay_set_noise": (_i, [_vp, _i, _vp, _i]),
    "sc_selfplay_get_noise": (_i, [_vp, _i, _vp, _i]),
    "sc_selfplay_set_position": (_i, [_vp, _i, _vp, _i]),
    "sc_encode_steps_last_timing": (_i, [C.POINTER(_f), C.POINTER(_f)]),
    "sc_encode_steps": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_trace_write_json": (_i, [C.c_char_p, C.POINTER(TraceInfo), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_move_uci": (_i, [C.c_uint16, C.c_char_p]),
    "sc_move_index": (_i, [C.c_uint16, _i]),
}

_lib = None


def lib_path():
    return _LIB_PATH


def lib():
    """Loads libsc_engine.so; raises EngineError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise EngineError(f"{_LIB_PATH} is missing: build it with `python smart-chess-rust_amd/build.py` "
                          "(there is no CPU fallback)")
    try:
        L = C.CDLL(_LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise EngineError(f"cannot load {_LIB_PATH}: {e}") from e
    for name, (res, args) in ABI.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        e = EngineError(f"libsc_engine error {rc}: {lib().sc_last_error().decode()}")
        e.code = rc
        raise e


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def move_uci(m):
    buf = C.create_string_buffer(8)
    lib().sc_move_uci(int(m), buf)
    return buf.value.decode()


def uci_move(s):
    promo = {"n": 2, "b": 3, "r": 4, "q": 5}
    f = (ord(s[1]) - 49) * 8 + ord(s[0]) - 97
    t = (ord(s[3]) - 49) * 8 + ord(s[2]) - 97
    return f | (t << 6) | ((promo[s[4]] if len(s) > 4 else 0) << 12)


class Engine:
    """The network backend (replaces ChessTS/ChessEP/ChessOnnx construction, src/main.rs:83-128)."""

    def __init__(self, n_res_blocks=10, channels=256, seed=0, weights=None, device=0, precision="bf16"):
        self.L = lib()
        self.n_res_blocks, self.channels = n_res_blocks, channels
        cfg = NetConfig(n_res_blocks, channels, seed, {"bf16": 0, "fp8": 1}[precision], 0)
        h = C.c_void_p()
        _check(self.L.sc_engine_create(C.byref(cfg), weights.encode() if weights else None, device, C.byref(h)))
        self.h = h
        self.precision = "fp8" if self.L.sc_engine_precision(h) == 1 else "bf16"   # an SCW2 blob decides by itself

    def close(self):
        if getattr(self, "h", None):
            self.L.sc_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, boards, meta, want_logp=True):
        """ChessModule.forward: boards int8[n,8,8,112], meta int32[n,7] -> logp[n,4672], value[n]"""
        boards = np.ascontiguousarray(boards, np.int8).reshape(-1, 8, 8, 112)
        meta = np.ascontiguousarray(meta, np.int32).reshape(-1, 7)
        n = boards.shape[0]
        logp = np.zeros((n, 4672), np.float32) if want_logp else None
        value = np.zeros(n, np.float32)
        _check(self.L.sc_forward_batch(self.h, n, _p(boards), _p(meta), _p(logp), _p(value)))
        return logp, value

    def debug(self, boards, meta, stage):
        boards = np.ascontiguousarray(boards, np.int8).reshape(-1, 8, 8, 112)
        meta = np.ascontiguousarray(meta, np.int32).reshape(-1, 7)
        n = boards.shape[0]
        out = np.zeros((n, 64, self.channels), np.float32)
        _check(self.L.sc_forward_debug(self.h, n, _p(boards), _p(meta), stage, _p(out)))
        return out

    def predict(self, boards, meta, legal_idx, argmax=False):
        """Game::predict tail: legal_idx = list (per position) of action indices -> (list of priors, value[n]);
        argmax: post_process_distr's one-hot branch (src/chess.rs:880-889)"""
        boards = np.ascontiguousarray(boards, np.int8).reshape(-1, 8, 8, 112)
        meta = np.ascontiguousarray(meta, np.int32).reshape(-1, 7)
        n = boards.shape[0]
        off = np.zeros(n + 1, np.uint32)
        off[1:] = np.cumsum([len(x) for x in legal_idx])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint16) for x in legal_idx]) if off[-1] else
                                    np.zeros(0, np.uint16), np.uint16)
        pri = np.zeros(int(off[-1]), np.float32)
        value = np.zeros(n, np.float32)
        fn = self.L.sc_predict_batch_argmax if argmax else self.L.sc_predict_batch
        _check(fn(self.h, n, _p(boards), _p(meta), _p(flat), _p(off), _p(pri), _p(value)))
        return [pri[off[i]:off[i + 1]] for i in range(n)], value


def encode_positions(move_lists, device=0, engine=None):
    """Rules + encoder on the GPU for positions given as move lists (uint16 moves or UCI strings)."""
    L = lib()
    n = len(move_lists)
    ml = [[uci_move(m) if isinstance(m, str) else int(m) for m in g] for g in move_lists]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum([len(g) for g in ml])
    flat = np.asarray([m for g in ml for m in g], np.uint16)
    if flat.size == 0:
        flat = np.zeros(1, np.uint16)
    boards = np.zeros((n, 8, 8, 112), np.int8)
    meta = np.zeros((n, 7), np.int32)
    lm = np.zeros((n, MAX_MOVES), np.uint16)
    li = np.zeros((n, MAX_MOVES), np.uint16)
    nl = np.zeros(n, np.int32)
    oc = np.zeros((n, 4), np.int32)
    _check(L.sc_encode_positions(engine.h if engine else None, device, n, _p(flat), _p(off), _p(boards), _p(meta), _p(lm),
                                 _p(li), _p(nl), _p(oc)))
    return dict(boards=boards, meta=meta, legal_moves=[lm[i, :nl[i]].copy() for i in range(n)],
                legal_idx=[li[i, :nl[i]].copy() for i in range(n)], n_legal=nl, termination=oc[:, 0], winner=oc[:, 1],
                is_check=oc[:, 2], status=oc[:, 3])


def encode_steps_batch(games, apply_mirror=False, device=0, engine=None):
    """Trace -> training tensors for a batch of games on the GPU (sc_encode_steps).
    games: list of step lists [(next_move, [(move, count), ...]), ...] with moves as uint16 or UCI strings -- the
    `steps` argument of libsmartchess.chess_encode_steps (reference src/lib.rs:46-50), one per game.
    -> dict(boards int8[P,8,8,112], meta int32[P,7], dist f32[P,4672], move_indices [P lists], ply_off[n+1], status[n])"""
    L = lib()
    mv = lambda m: uci_move(m) if isinstance(m, str) else int(m)
    n = len(games)
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum([len(g) for g in games])
    P = int(off[n])
    flat = np.asarray([mv(s[0]) for g in games for s in g] or [0], np.uint16)
    coff = np.zeros(P + 1, np.uint32)
    coff[1:] = np.cumsum([len(s[1]) for g in games for s in g])
    cm = np.asarray([mv(c[0]) for g in games for s in g for c in s[1]] or [0], np.uint16)
    cn = np.asarray([int(c[1]) for g in games for s in g for c in s[1]] or [0], np.uint32)
    boards = np.zeros((max(P, 1), 8, 8, 112), np.int8)
    meta = np.zeros((max(P, 1), 7), np.int32)
    dist = np.zeros((max(P, 1), 4672), np.float32)
    li = np.zeros((max(P, 1), MAX_MOVES), np.uint16)
    nl = np.zeros(max(P, 1), np.int32)
    status = np.zeros(max(n, 1), np.int32)
    _check(L.sc_encode_steps(engine.h if engine else None, device, n, _p(flat), _p(off), _p(cm), _p(cn), _p(coff),
                             int(bool(apply_mirror)), _p(boards), _p(meta), _p(dist), _p(li), _p(nl), _p(status)))
    return dict(boards=boards[:P], meta=meta[:P], dist=dist[:P], move_indices=[li[i, :nl[i]].astype(np.int32) for i in range(P)],
                ply_off=off, status=status[:n])


def encode_steps_last_timing():
    """(kernel ms, whole-call ms) of this thread's last sc_encode_steps"""
    a, b = C.c_float(0), C.c_float(0)
    lib().sc_encode_steps_last_timing(C.byref(a), C.byref(b))
    return a.value, b.value


def encode_steps(steps, apply_mirror=False, device=0, engine=None):
    """Mirror of libsmartchess.chess_encode_steps(steps, apply_mirror) (reference src/lib.rs:46-128, used by
    py/dataset.py:77): one game -> [(boards int8[8,8,112], meta int32[7], dist f32[4672], move_indices), ...].
    Raises EngineError where the reference panics (children != legal moves, or an illegal played move)."""
    r = encode_steps_batch([steps], apply_mirror, device, engine)
    st = int(r["status"][0])
    if st >= 1000:
        raise EngineError(f"inconsistent moves at ply {st - 1000}")
    if st < 0:
        raise EngineError(f"num_act table doesn't include the next move (ply {-st - 1})")
    return [(r["boards"][i], r["meta"][i], r["dist"][i], r["move_indices"][i]) for i in range(len(steps))]


class Play:
    """Interactive engine handle: the `chess_play_*` functions of the reference's Python extension
    (src/lib.rs:161-358: new / mcts / step / apply_move / inspect / dump_search_tree / inference / encode) on one
    search slot of the GPU engine.  Differences: moves are UCI strings; `inspect()` returns the move list instead of a
    python-chess board object; the tree keeps only the current subtree (the reference also keeps the never revisited
    siblings of played moves), so `dump_search_tree()` shows the played line as a chain of single children."""

    def __init__(self, engine, initial_moves=(), evaluator="net", seed=0):
        self.engine = engine
        self.moves = [m if isinstance(m, str) else move_uci(m) for m in initial_moves]
        self._seed = seed
        # rollout_num is the per-ply budget of the self-play driver: huge here, plies advance only through step()
        self.sp = SelfPlay(engine, n_slots=1, n_games=1, rollout_num=60000, num_steps=4000, with_noise=False, outcome_gate=1 << 30,
                           evaluator=evaluator, seed=seed)
        self._rng = np.random.default_rng(seed)
        self._set()

    def _set(self):
        mv = np.asarray([uci_move(m) for m in self.moves] or [0], np.uint16)
        _check(self.sp.L.sc_selfplay_set_position(self.sp.h, 0, _p(mv), len(self.moves)))

    def close(self):
        self.sp.close()

    def mcts(self, rollout, cpuct=2.5, noise=False):
        """chess_play_mcts: `rollout` more simulations on the current tree (epsilon 0.15 as lib.rs:243)"""
        _check(self.sp.L.sc_selfplay_set_search(self.sp.h, cpuct, 0.15, int(bool(noise))))
        self.sp.enqueue(rollout)
        self.sp.sync()

    def _root_children(self):
        t = self.sp.tree(0)
        if t["n"].size == 0 or t["n_child"][0] == 0:
            return t, 0, 0
        return t, int(t["first_child"][0]), int(t["n_child"][0])

    def step(self, temp=0.0):
        """chess_play_step = mcts::step (src/mcts.rs:292-328): temperature 0 -> first most-visited child, else a
        sample ~ N^(1/temp); descends and starts a fresh tree there.  Returns the move or None (no children)."""
        t, fc, nc = self._root_children()
        if nc == 0:
            return None
        n = t["n"][fc:fc + nc].astype(np.float32)
        if temp == 0.0:
            choice = int(np.argmax(n))
        else:
            w = n ** np.float32(1.0 / temp)
            choice = int(self._rng.choice(nc, p=(w / w.sum()).astype(np.float64)))
        mv = move_uci(t["move"][fc + choice])
        self.apply_move(mv)
        return mv

    def apply_move(self, mov):
        """chess_play_apply_move: play `mov` and continue from a fresh node"""
        self.moves.append(mov if isinstance(mov, str) else move_uci(mov))
        self._set()

    def inspect(self):
        """chess_play_inspect -> (None, move stack newest first, q_value of the current node, [(move, N, Q), ...])"""
        t, fc, nc = self._root_children()
        q = float(t["q"][0]) if t["q"].size else 0.0
        ch = [(move_uci(t["move"][fc + i]), int(t["n"][fc + i]), float(t["q"][fc + i])) for i in range(nc)]
        return None, list(reversed(self.moves)), q, ch

    def dump_search_tree(self):
        """chess_play_dump_search_tree: nested dicts with serde's field names (src/mcts.rs:43-56: step, depth, q,
        num_act, children); step = [uci or None, colour of the side to move at the node]"""
        t = self.sp.tree(0)
        d0 = len(self.moves)

        def node(i, depth, mv):
            colour = "White" if depth % 2 == 0 else "Black"
            fc, nc = int(t["first_child"][i]), int(t["n_child"][i])
            kids = [node(fc + k, depth + 1, move_uci(t["move"][fc + k])) for k in range(nc)] if fc >= 0 else []
            return {"step": [mv, colour], "depth": depth, "q": float(t["q"][i]), "num_act": int(t["n"][i]), "children": kids}
        cur = node(0, d0, self.moves[-1] if self.moves else None) if t["n"].size else None
        for d in range(d0 - 1, -1, -1):   # the played line above the current node
            cur = {"step": [self.moves[d - 1] if d > 0 else None, "White" if d % 2 == 0 else "Black"], "depth": d, "q": 0.0,
                   "num_act": 0, "children": [cur]}
        return cur

    def inference(self):
        """chess_play_inference -> (legal moves, priors, value) of the current position (Game::predict)"""
        steps, pri, val = ChessHip(self.engine).predict(self.moves)
        return [move_uci(m) for m in steps], pri, val

    def encode(self):
        """chess_play_encode -> (boards int8[8,8,112], meta int32[7])"""
        e = encode_positions([self.moves], engine=self.engine)
        return e["boards"][0], e["meta"][0]


def encode_move(turn_white, move):
    """libsmartchess.chess_encode_move(turn, move) (reference src/lib.rs:37-44); no GPU needed"""
    return int(lib().sc_move_index(uci_move(move) if isinstance(move, str) else int(move), int(bool(turn_white))))


def search(engine, moves, rollout, cpuct=2.5, noise=False, seed=0):
    """sc_search: one search from the position after `moves` -> (root_q, [(uci, N, Q, prior), ...])"""
    mv = np.asarray([uci_move(m) if isinstance(m, str) else int(m) for m in moves] or [0], np.uint16)
    cm, cn = np.zeros(MAX_MOVES, np.uint16), np.zeros(MAX_MOVES, np.int32)
    cq, cp = np.zeros(MAX_MOVES, np.float32), np.zeros(MAX_MOVES, np.float32)
    rq = C.c_float(0)
    n = lib().sc_search(engine.h, _p(mv), len(moves), rollout, cpuct, int(bool(noise)), seed, MAX_MOVES, _p(cm), _p(cn), _p(cq), _p(cp),
                        C.byref(rq))
    if n < 0:
        _check(n)
    return rq.value, [(move_uci(cm[i]), int(cn[i]), float(cq[i]), float(cp[i])) for i in range(n)]


def elo(total, wins, losses):
    """scripts/elo.py:15-21: Elo difference from Total/Win/Lost"""
    import math
    s = (wins + (total - wins - losses) / 2) / total
    if s <= 0.0 or s >= 1.0:
        return math.copysign(math.inf, s - 0.5)
    return 400 * math.log(s / (1 - s), 10)


def play_match(a, b, n_games=100, rollout=100, cpuct=1.5, temperature=0.0, temperature_switch=0, num_steps=200, seed=0,
               swap=True):
    """Batched `scripts/leader-board:44-54`: n_games with engine `a` as White and `b` as Black, then (swap) the same
    number with the colours exchanged; every game is `play`'s loop (src/play.rs:241-343: no noise, outcome after every
    ply, at most 200 plies, random tie-break).  -> dict(results per colour assignment, a's score, Elo of a over b)."""
    out = {"as_white": None, "as_black": None}
    tot = win = lost = 0
    # Both colour assignments play at the same time: each handle launches on the stream of its White engine, so the two sets of
    # n_games workgroups share the GPU (100 + 100 of 256 CUs for the reference's 100-game matches) instead of running one after the
    # other.  A game depends only on its seed and id: the results are those of the sequential loop.
    handles = []
    for key, (w, bl) in (("as_white", (a, b)), ("as_black", (b, a))):
        if key == "as_black" and not swap:
            break
        sp = SelfPlay(w, n_slots=n_games, n_games=n_games, rollout_num=rollout, num_steps=num_steps, cpuct=cpuct,
                      temperature=temperature, temperature_switch=temperature_switch, with_noise=False, outcome_gate=-1,
                      seed=seed + (0 if key == "as_white" else 1), tie_random=True)
        sp.set_players(w, bl)
        handles.append((key, sp))
    live = [sp for _, sp in handles]
    while live:
        for _ in range(2):                       # two plies per look at the statistics
            if len(live) > 1:
                enqueue_interleaved(live, rollout)
            else:
                live[0].enqueue(rollout)
        live = [sp for sp in live if sp.stats()["games_active"] > 0]
    for key, sp in handles:
        res = {"White": 0, "Black": 0, "draw": 0, "unfinished": 0}
        traces = []
        for g in range(n_games):
            t = sp.trace(g)
            traces.append(t)
            oc = t["outcome"] if t else None
            if oc is None:
                res["unfinished"] += 1
            elif oc["winner"] is None:
                res["draw"] += 1
            else:
                res[oc["winner"]] += 1
        sp.close()
        out[key] = dict(results=res, traces=traces)
        a_col, b_col = ("White", "Black") if key == "as_white" else ("Black", "White")
        tot += n_games
        win += res[a_col]
        lost += res[b_col]
    out.update(total=tot, a_wins=win, b_wins=lost, elo_a_minus_b=elo(tot, win, lost))
    return out


class ChessHip:
    """Mirror of `impl Game<BoardState> for ChessTS` (src/backends/torch.rs:34-53) over the GPU engine.

    A node/state pair of the reference is represented by the list of moves played from the start
    position (that is what `_encode` reconstructs from the tree's parent chain and the board's
    move stack, src/chess.rs:845-867).
    """

    def __init__(self, engine):
        self.engine = engine

    def predict(self, moves, argmax=False):
        """-> (steps, priors, value): steps = legal moves (uint16) in python-chess order; empty at game end,
        with value = +1 White won / -1 Black won / 0 (torch.rs:98-106)."""
        enc = encode_positions([moves], engine=self.engine)
        if enc["status"][0] < 0:
            raise EngineError(f"illegal move at index {-enc['status'][0] - 1}")
        if enc["n_legal"][0] == 0:
            w = enc["winner"][0]
            return [], np.zeros(0, np.float32), (1.0 if w == 1 else -1.0 if w == 0 else 0.0)
        pri, val = self.engine.predict(enc["boards"], enc["meta"], [enc["legal_idx"][0]], argmax=argmax)
        return list(enc["legal_moves"][0]), pri[0], float(val[0])

    @staticmethod
    def reverse_q(moves):
        """node.step.1 == Black (torch.rs:49-52): Black is to move after an odd number of plies"""
        return len(moves) % 2 == 1


class SelfPlay:
    """Batched `selfplay` (src/main.rs): same option names as the reference CLI (main.rs:25-60)."""

    def __init__(self, engine=None, n_slots=256, n_games=None, rollout_num=180, num_steps=150, cpuct=2.5,
                 temperature=0.0, temperature_switch=4, epsilon=0.15, with_noise=True, outcome_gate=100,
                 evaluator="net", external_noise=False, seed=0, first_game_id=0, trace_capacity=0, own_stream=False, device=0,
                 tie_random=False, trace_hold=False, rollout_factor=0.0):
        self.L = lib()
        self.engine = engine
        cfg = SelfplayConfig(n_slots, n_games if n_games is not None else n_slots, rollout_num, num_steps, cpuct,
                             temperature, temperature_switch, epsilon, int(with_noise), outcome_gate,
                             EVALUATORS[evaluator], int(external_noise), seed, first_game_id, trace_capacity, int(own_stream),
                             int(tie_random), int(trace_hold), float(rollout_factor))
        self.cfg = cfg
        h = C.c_void_p()
        _check(self.L.sc_selfplay_create(engine.h if engine else None, device, C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.sc_selfplay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enqueue(self, n_sims):
        _check(self.L.sc_selfplay_enqueue_sims(self.h, n_sims))

    def set_players(self, white=None, black=None, salt_white=0, salt_black=0):
        """match play (src/play.rs:318-343): even plies are searched by `white`, odd plies by `black`"""
        self._players = (white, black)   # keep the engines alive
        _check(self.L.sc_selfplay_set_players(self.h, white.h if white else None, black.h if black else None, salt_white, salt_black))

    def sync(self):
        _check(self.L.sc_selfplay_synchronize(self.h))

    def run(self, max_sim_steps=0):
        _check(self.L.sc_selfplay_run(self.h, max_sim_steps))

    def stats(self):
        s = Stats()
        _check(self.L.sc_selfplay_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def enable_timing(self, stride=1):
        _check(self.L.sc_selfplay_enable_timing(self.h, stride))

    def launches_per_step(self):
        """1: fused step kernel with the value FC inside, 2: fused step kernel + value FC launch, 3: separate launches"""
        return int(self.L.sc_selfplay_launches_per_step(self.h))

    def timing(self, reset=True):
        a, b, n = C.c_float(0), C.c_float(0), C.c_int64(0)
        _check(self.L.sc_selfplay_timing(self.h, int(reset), C.byref(a), C.byref(b), C.byref(n)))
        return dict(ms_total=a.value, ms_tower_sum=b.value, tower_launches=n.value)

    def poll(self, cap=4096):
        """sc_selfplay_poll: handle-local indices of the games that finished since they were last reported"""
        buf = np.zeros(max(cap, 1), np.int32)
        n = self.L.sc_selfplay_poll(self.h, _p(buf), cap)
        if n < 0:
            _check(n)
        return [int(x) for x in buf[:n]]

    def trace(self, game):
        """-> dict in the reference's trace-file shape (src/trace.rs:5-9); None if unfinished; raises EngineError when the
        trace has left the device (ring row overwritten or released)"""
        info = TraceInfo()
        rc = self.L.sc_selfplay_get_trace(self.h, game, C.byref(info), None, None, None, None, None, None, None)
        if rc == 1:
            return None
        _check(rc)
        ns, nc = info.n_steps, info.n_children_total
        sm, sq = np.zeros(ns + 1, np.uint16), np.zeros(ns + 1, np.float32)
        co = np.zeros(ns + 2, np.int32)
        cm, cn = np.zeros(nc + 1, np.uint16), np.zeros(nc + 1, np.int32)
        cq, cu = np.zeros(nc + 1, np.float32), np.zeros(nc + 1, np.float32)
        _check(self.L.sc_selfplay_get_trace(self.h, game, C.byref(info), _p(sm), _p(sq), _p(co), _p(cm), _p(cn), _p(cq),
                                            _p(cu)))
        steps = []
        for i in range(ns):
            ch = [(move_uci(cm[j]), int(cn[j]), float(cq[j]), float(cu[j])) for j in range(co[i], co[i + 1])]
            steps.append((move_uci(sm[i]), float(sq[i]), ch))
        outcome = None
        if info.has_outcome:
            outcome = {"termination": TERMINATION[info.termination], "winner": _WINNER[info.winner]}
        return {"steps": steps, "outcome": outcome, "game_id": int(info.game_id)}

    def write_trace(self, game, path):
        _check(self.L.sc_selfplay_write_trace_json(self.h, game, path.encode()))

    def stream_traces(self, path_of, chunk=None):
        """The loop of lib/sc-selfplay (src/main.rs:235-238 writes each game's file when it ends): plays every game of the
        handle and writes `path_of(game_id)` as games finish, from a handle created with a trace ring and trace_hold=True.
        The next ply's simulation steps are enqueued BEFORE the finished games' traces are fetched and written: a reported
        row is final, it is read while the GPU searches.  -> number of files written."""
        chunk = chunk or self.cfg.rollout_num
        written = 0
        self.enqueue(chunk)
        while True:
            fin = self.poll()
            active = self.stats()["games_active"]
            if active:
                self.enqueue(chunk)
            for g in fin:
                self.write_trace(g, path_of(self.cfg.first_game_id + g))
                written += 1
            if not active and not fin:
                return written

    def tree(self, slot, cap=1 << 20):
        n = self.L.sc_selfplay_get_tree(self.h, slot, 0, None, None, None, None, None, None, None)
        if n < 0:
            _check(n)
        n = min(n, cap)
        out = dict(n=np.zeros(n, np.int32), q=np.zeros(n, np.float32), uct=np.zeros(n, np.float32),
                   prior=np.zeros(n, np.float32), move=np.zeros(n, np.uint16), first_child=np.zeros(n, np.int32),
                   n_child=np.zeros(n, np.int32))
        r = self.L.sc_selfplay_get_tree(self.h, slot, n, _p(out["n"]), _p(out["q"]), _p(out["uct"]), _p(out["prior"]),
                                        _p(out["move"]), _p(out["first_child"]), _p(out["n_child"]))
        if r < 0:
            _check(r)
        return out

    def debug_cycles(self, enable=True, read=False):
        """sc_selfplay_debug_cycles: switch the stamps on / read those of the last launch -> uint64[n_slots, 32] (or None)"""
        out = np.zeros((self.cfg.n_slots, 32), np.uint64) if read else None
        _check(self.L.sc_selfplay_debug_cycles(self.h, int(enable), _p(out)))
        return out

    def slot(self, slot):
        ply, sim, st, plen = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        gid = C.c_uint64(0)
        path = np.zeros(1024, np.int32)
        _check(self.L.sc_selfplay_get_slot(self.h, slot, C.byref(ply), C.byref(sim), C.byref(st), C.byref(gid), _p(path),
                                           C.byref(plen)))
        return dict(ply=ply.value, sim=sim.value, status=st.value, game_id=gid.value, path=path[:plen.value].copy())

    def set_noise(self, slot, noise):
        noise = np.ascontiguousarray(noise, np.float32)
        _check(self.L.sc_selfplay_set_noise(self.h, slot, _p(noise), noise.size))

    def get_noise(self, slot, n):
        out = np.zeros(MAX_MOVES, np.float32)
        _check(self.L.sc_selfplay_get_noise(self.h, slot, _p(out), MAX_MOVES))
        return out[:n]

    def set_position(self, slot, moves):
        mv = np.asarray([uci_move(m) if isinstance(m, str) else int(m) for m in moves], np.uint16)
        if mv.size == 0:
            mv = np.zeros(1, np.uint16)
        _check(self.L.sc_selfplay_set_position(self.h, slot, _p(mv), len(moves)))


def find_max(values, device=0):
    """sc_debug_find_max: (one-round result or -2, four-round result) of the descent's arg-max on `values`"""
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros(2, np.int32)
    _check(lib().sc_debug_find_max(device, _p(v), v.size, _p(out)))
    return int(out[0]), int(out[1])


def runtime_flags():
    return int(lib().sc_runtime_flags())


def enqueue_interleaved(handles, n_sims):
    """n simulation steps on several SelfPlay handles (own_stream=True), interleaved step by step"""
    arr = (C.c_void_p * len(handles))(*[h.h for h in handles])
    _check(lib().sc_selfplay_enqueue_interleaved(arr, len(handles), n_sims))


def write_trace_json(path, trace):
    """sc_trace_write_json on a trace dict (no GPU needed)."""
    steps = trace["steps"]
    ns = len(steps)
    info = TraceInfo()
    info.n_steps = ns
    oc = trace.get("outcome")
    info.has_outcome = int(oc is not None)
    inv_t = {v: k for k, v in TERMINATION.items()}
    info.termination = inv_t[oc["termination"]] if oc else 0
    info.winner = {"White": 1, "Black": 0, None: -1}[oc["winner"]] if oc else -1
    sm = np.asarray([uci_move(s[0]) for s in steps] + [0], np.uint16)
    sq = np.asarray([s[1] for s in steps] + [0], np.float32)
    co = np.zeros(ns + 2, np.int32)
    co[1:ns + 1] = np.cumsum([len(s[2]) for s in steps])
    ch = [c for s in steps for c in s[2]]
    info.n_children_total = len(ch)
    cm = np.asarray([uci_move(c[0]) for c in ch] + [0], np.uint16)
    cn = np.asarray([c[1] for c in ch] + [0], np.int32)
    cq = np.asarray([c[2] for c in ch] + [0], np.float32)
    cu = np.asarray([c[3] for c in ch] + [0], np.float32)
    _check(lib().sc_trace_write_json(path.encode(), C.byref(info), _p(sm), _p(sq), _p(co), _p(cm), _p(cn), _p(cq), _p(cu)))
