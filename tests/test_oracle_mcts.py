"""The oracle's restatement of src/mcts.rs / src/main.rs: arithmetic and tie rules against hand-computed
values, faithful (re-predict at every node, mcts.rs:152) == cached-prior mode, determinism."""
import ctypes as C

import numpy as np


def _uct_np(sqrt_total, prior, q, n, reverse, cpuct):
    f = np.float32
    avg = f(f(q) / f(f(n) + f(1e-4))) * (f(-1) if reverse else f(1))
    expl = f(f(f(f(sqrt_total) + f(0.01)) / f(f(1) + f(n))) * f(cpuct)) * f(prior)
    return f(avg + expl)


def test_faithful_equals_cached(orc):
    st = orc.State()
    for m in ["e2e4", "e7e5", "g1f3"]:
        st.push(m)
    a, b = orc.Search(st), orc.Search(st)
    for _ in range(150):
        a.sim(faithful=True)
        b.sim(faithful=False)
    da, db = a.dump(), b.dump()
    assert all(np.array_equal(da[k], db[k]) for k in da)
    assert a.num_evals() > b.num_evals() == 150      # the reference's cost: one net call per path node


def test_uct_arithmetic_and_invariants(orc):
    st = orc.State()
    s = orc.Search(st)
    R = 60
    for i in range(R):
        s.sim(cpuct=2.5)
    d = s.dump()
    assert d["n"][0] == R and d["n_child"][0] == 20
    kids = slice(d["first_child"][0], d["first_child"][0] + 20)
    assert d["n"][kids].sum() == R - 1               # first simulation expands the root
    # recompute the uct stored on the root's children at the LAST selection (they were written before
    # the last backup): rebuild from the previous visit counts is not possible, so check a fresh search
    s2 = orc.Search(st)
    s2.sim(); s2.sim()
    d2 = s2.dump()
    # second simulation: all children N=0,Q=0 -> uct = (sqrt(0)+0.01)/(1+0)*cpuct*prior; last max wins ties
    pri = np.zeros(256, np.float32); val = C.c_float(0)
    L = orc.lib()
    lm = st.legal_moves()
    legal = (C.c_uint16 * len(lm))(*lm); idx = (C.c_int * len(lm))()
    L.orc_eval_synth.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 4
    L.orc_eval_synth(None, st.h, len(lm), legal, idx, pri.ctypes.data, C.byref(val))
    want = np.array([_uct_np(0.0, pri[i], 0.0, 0, False, 2.5) for i in range(20)], np.float32)
    assert np.array_equal(d2["uct"][1:21], want)
    chosen = int(np.nonzero(d2["n"][1:21])[0][0])
    assert chosen == int(np.argmax(want))            # the hash evaluator's priors never collide: no tie here (see below)


def test_find_max_keeps_the_last_maximum(orc):
    """find_max = Iterator::max_by (src/mcts.rs:78-88): of equal PUCT values the LAST child wins.  Hand-computed on the
    uniform test evaluator (prior 1/20 for every root move, value 0):
      sim 1 expands the root; sim 2: all 20 children have u = (0 + 0.01) / 1 * 2.5 * 0.05 -> child 19 (node 20);
      sim 3: child 19 has N = 1, u = (1 + 0.01) / 2 * 2.5 * 0.05 < the others' (1 + 0.01) / 1 * 2.5 * 0.05 -> child 18;
      ... sims 2..21 visit the children in the order 19, 18, .., 0; sim 22: all N = 1 tie again -> child 19, and below it
      the last of ITS children."""
    f = np.float32
    s = orc.Search(orc.State())
    s.sim(evaluator="orc_eval_synth_uniform", cpuct=2.5)
    assert list(s.last_path()) == [0]
    for k in range(2, 22):
        s.sim(evaluator="orc_eval_synth_uniform", cpuct=2.5)
        assert list(s.last_path()) == [0, 22 - k], k
        d = s.dump()
        tot = f(np.sqrt(f(k - 2)))
        u_unvisited = f(f(f(tot + f(0.01)) / f(1)) * f(2.5)) * f(f(1) / f(20))
        assert d["uct"][22 - k] == u_unvisited                       # value written at selection time
    s.sim(evaluator="orc_eval_synth_uniform", cpuct=2.5)
    d = s.dump()
    fc, nc = int(d["first_child"][20]), int(d["n_child"][20])
    assert nc > 1 and list(s.last_path()) == [0, 20, fc + nc - 1]
    assert np.all(d["uct"][1:21] == d["uct"][1])                     # an exact 20-way tie, resolved to the last
    # coarse evaluator: ties between SOME siblings (equal 2-bit weights) next to non-zero value sums
    c = orc.Search(orc.State())
    for _ in range(2):
        c.sim(evaluator="orc_eval_synth_coarse", cpuct=2.5)
    d = c.dump()
    u = d["uct"][1:21]
    assert (u == u.max()).sum() > 1 and int(c.last_path()[1]) - 1 == max(i for i in range(20) if u[i] == u.max())


def test_rollout_factor_budget(orc):
    """--rollout-factor (src/main.rs:175-176): min(300, (n_legal as f32 * v) as i32) simulations per ply"""
    g = orc.selfplay_game(rollout_num=300, num_steps=6, with_noise=False, seed=2, rollout_factor=1.5)
    st = orc.State()
    for mv, q, kids in g["steps"]:
        n = len(st.legal_moves())
        assert len(kids) == n and sum(k[1] for k in kids) == min(300, int(np.float32(n) * np.float32(1.5))) - 1
        st.push(mv)
    assert orc.selfplay_game(rollout_num=300, num_steps=2, with_noise=False, seed=2, rollout_factor=40.0)["n_sims"] == 600


def test_choose_child_rules(orc):
    L = orc.lib()
    n = np.array([3, 7, 7, 1], np.int32)
    assert L.orc_choose_child(n.ctypes.data, 4, 0.0, 0.5) == 1          # temp 0: FIRST max (mcts.rs:309-311)
    # temp 1: cumulative 3,10,17,(18): x = u*18
    assert L.orc_choose_child(n.ctypes.data, 4, 1.0, 0.0) == 0
    assert L.orc_choose_child(n.ctypes.data, 4, 1.0, 3.0 / 18 + 1e-6) == 1
    assert L.orc_choose_child(n.ctypes.data, 4, 1.0, 0.999) == 3
    assert L.orc_choose_child(n.ctypes.data, 0, 1.0, 0.5) == -1         # no children -> None


def test_selfplay_trace_semantics(orc):
    g = orc.selfplay_game(rollout_num=30, num_steps=150, with_noise=True, seed=4, game_id=2)
    assert g == orc.selfplay_game(rollout_num=30, num_steps=150, with_noise=True, seed=4, game_id=2)
    st = orc.State()
    for mv, q_root, kids in g["steps"]:
        assert [k[0] for k in kids] == st.legal_uci()
        assert sum(k[1] for k in kids) == 30 - 1
        assert mv in [k[0] for k in kids]
        st.push(mv)
    if len(g["steps"]) < 150:
        assert g["outcome"] is not None and st.outcome() == g["outcome"]
    else:
        assert g["outcome"] is None                                      # hitting --num-steps leaves outcome null
    # terminal positions back up +-1: checkmate for White seen from White = +1
    g2 = orc.selfplay_game(rollout_num=30, num_steps=8, with_noise=False, seed=4)
    assert len(g2["steps"]) == 8 and g2["outcome"] is None


def test_match_game_rules(orc):
    """oracle restatement of the `play` loop (src/play.rs:241-343): players alternate by ply, outcome after every ply,
    temperature-0 ties resolved among the most visited children only"""
    cfg = dict(rollout_num=12, num_steps=40, cpuct=1.5, temperature=0.0, temperature_switch=0, seed=4, game_id=7)
    t = orc.match_game(user_white=1, user_black=2, **cfg)
    assert t == orc.match_game(user_white=1, user_black=2, **cfg)
    assert t["steps"] != orc.match_game(user_white=2, user_black=1, **cfg)["steps"]
    st = orc.State()
    for i, s in enumerate(t["steps"]):
        assert sorted(c[0] for c in s[2]) == sorted(st.legal_uci())
        mx = max(c[1] for c in s[2])
        assert dict((c[0], c[1]) for c in s[2])[s[0]] == mx          # a most-visited child was played
        assert sum(c[1] for c in s[2]) == cfg["rollout_num"] - 1     # fresh tree every ply (step() resets the child)
        st.push(orc.from_uci(s[0]))
        assert (st.outcome() is not None) == (t["outcome"] is not None and i == len(t["steps"]) - 1)
    # identical players and no tie => the same game as noise-free self-play consulted every ply
    same = orc.match_game(user_white=None, user_black=None, **cfg)
    sp = orc.selfplay_game(rollout_num=12, num_steps=40, cpuct=1.5, temperature=0.0, temperature_switch=0, with_noise=False,
                           seed=4, game_id=7, outcome_gate=-1)
    k = next((i for i, s in enumerate(same["steps"]) if sum(c[1] == max(x[1] for x in s[2]) for c in s[2]) > 1), len(same["steps"]))
    assert same["steps"][:k] == sp["steps"][:k]
