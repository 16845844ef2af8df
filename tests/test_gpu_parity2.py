"""GPU parity tests, part 2 (-m gpu): the device code paths the first file never reached -- exact PUCT ties and the
last-maximum rule (src/mcts.rs:78-88), nodes with more than 64 children (the four-round arg-max of the descent),
BASELINE configs[1] and configs[3] at full size, the reference's 19/20-block goldens, --rollout-factor, the trace ring
and the streaming drain.  Everything goes through the C ABI (include/sc_engine.h)."""
import json
import os

import numpy as np
import pytest

from helpers import random_games

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = ATOL = 1e-2   # reference convention, scripts/eval_speed.py:40-43

# 18 plies from the start position to a White-to-move position with 82 legal moves (found by a beam search on the
# oracle's move generator, tools/find_wide_position.py): the root of a search from here has more than 64 children
WIDE = ("e2e3 d7d5 d1g4 d8d6 f1b5 e8d8 b1c3 d6h2 c3d5 h2d6 h1h6 d6a3 b2b3 a3a4 c1b2 c7c5 b2e5 a4b3").split()
# 112 plies to a White-to-move position with 137 legal moves (seven white queens; found by a promotion-driven greedy search on
# the oracle's move generator): a root with more than 128 children -- the third round of lanes of the descent's wide level
WIDE137 = ("h2h4 g8h6 a2a4 h6g4 f2f4 g4h6 a4a5 h6g4 d2d4 g4f2 e1f2 b8a6 h4h5 a6b4 c2c4 b4c2 "
           "d1c2 d7d5 c4c5 e8d7 f4f5 d7e8 c2a4 b7b5 a5b6 c7c6 a4c6 c8d7 c6d5 f7f6 b6b7 d8c8 "
           "c5c6 e8d8 d5a5 d8e8 c6d7 e8f7 a5a2 c8c4 a2c4 e7e6 b7a8q f7g8 d7d8q g8f7 f5e6 f7g8 "
           "g2g4 a7a5 d4d5 g7g6 h5h6 f6f5 g4g5 a5a4 e6e7 g8f7 e7e8q f7g8 b2b4 f5f4 e2e4 f4f3 "
           "e4e5 a4a3 e5e6 a3a2 e6e7 a2b1q a1a6 b1f5 d5d6 f5f7 d6d7 f7d5 a6g6 h7g6 e7f8q g8h7 "
           "d8e7 d5f7 d7d8q h8g8 e8f7 h7h8 h6h7 g8f8 b4b5 f8g8 b5b6 g8e8 b6b7 e8g8 b7b8q g8e8 "
           "h1h2 e8g8 c4c2 g8e8 e7e4 e8g8 a8a3 g8e8 f7e8 h8g7 f2g3 f3f2 e8e5 g7f7 h7h8q f2g1b").split()
MATED = ["f2f3", "e7e5", "g2g4", "d8h4"]   # White is checkmated: a game set to this line ends after its first search


@pytest.fixture(scope="module")
def scamd():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
    import scamd as m
    if m.lib().sc_device_count() <= 0:
        pytest.fail("no MI355X visible: the HIP path cannot be tested (and there is no fallback)")
    return m


def _same_tree(t, d):
    return (len(t["n"]) == len(d["n"]) and np.array_equal(t["n"], d["n"]) and np.array_equal(t["q"], d["q"])
            and np.array_equal(t["uct"], d["uct"]) and np.array_equal(t["move"][1:], d["move"][1:])
            and np.array_equal(t["n_child"], d["n_child"]))


def _pushed(orc, moves):
    st = orc.State()
    for m in moves:
        st.push(m)
    return st


# ---------------------------------------------------------------------------------- find_max (a3)
def test_find_max_device_forms(scamd):
    """both arg-max forms of the descent on crafted values: Iterator::max_by keeps the LAST maximum"""
    def last_max(u):
        return max(i for i in range(len(u)) if u[i] == max(u))   # == on floats: -0.0 == +0.0, as PartialOrd compares

    f = np.float32
    cases = [[0.5] * 20, [0.5] * 64, [0.5] * 65, [0.5] * 218, [0.5] * 256, [1.0, 2.0, 2.0, 1.0], [-1.0, -3.0, -1.0, -2.0],
             [f(0.0), f(-0.0)], [f(-0.0), f(0.0), f(-1.0)], [f(-0.0)] * 7 + [f(-2.0)], [3.0] + [1.0] * 63, [1.0] * 63 + [3.0],
             [1.0] * 64 + [3.0] + [1.0] * 100, [1.0] * 130 + [3.0, 3.0] + [1.0] * 50, [3.0] + [1.0] * 200 + [3.0] + [2.0] * 10,
             [-5.0] * 100 + [-4.0] + [-5.0] * 99 + [-4.0], [1e-30, 1e-30, 0.0], [-1e-30, -1e-30, -1.0],
             [f(0.12625)] * 19 + [f(0.063125)]]
    rnd = np.random.RandomState(5)
    for n in (2, 17, 63, 64, 65, 127, 128, 129, 191, 192, 193, 218, 255, 256):
        cases.append(list(rnd.randint(-3, 4, n).astype(np.float32) * f(0.25)))       # many exact ties, both signs
        cases.append(list(rnd.standard_normal(n).astype(np.float32)))
    for u in cases:
        one, four = scamd.find_max(u)
        want = last_max([float(x) for x in u])
        assert four == want, (len(u), u[:8])
        assert one == (want if len(u) <= 64 else -2), (len(u), u[:8])


# ---------------------------------------------------------------------------------- ties in the search (a3, a5)
@pytest.mark.parametrize("evaluator,oeval", [("synth_uniform", "orc_eval_synth_uniform"), ("synth_coarse", "orc_eval_synth_coarse")])
@pytest.mark.parametrize("line", [[], WIDE], ids=["start20", "wide82"])
def test_search_with_exact_ties_lockstep(scamd, orc, evaluator, oeval, line):
    """uniform priors + value 0: every unvisited sibling ties and the device must descend into the LAST child, like
    find_max (src/mcts.rs:78-88); coarse priors/values: ties between some siblings next to non-zero value sums.
    Node pools, uct words and paths equal the oracle after every simulation -- with <= 64 and with > 64 children."""
    R = 200
    sp = scamd.SelfPlay(None, n_slots=1, n_games=1, rollout_num=R, num_steps=30, cpuct=2.5, with_noise=False, evaluator=evaluator, seed=3)
    st = _pushed(orc, line)
    n_root = len(st.legal_moves())
    assert n_root == (82 if line else 20)
    sp.set_position(0, line)
    srch = orc.Search(st)
    for s in range(R - 1):
        sp.enqueue(1)
        srch.sim(evaluator=oeval, cpuct=2.5, with_noise=False)
        path = list(sp.slot(0)["path"])
        assert path == list(srch.last_path()), s
        if evaluator == "synth_uniform" and 1 <= s <= (n_root if not line else 60):
            # children are visited last to first while they all tie (in the wide position a mating move is among the
            # 82: its +1 reward ends the pure tie pattern once it has been visited)
            assert path[1] == n_root + 1 - s, s
        if s % 5 == 0 or s > R - 4:
            assert _same_tree(sp.tree(0), srch.dump()), s
    t = sp.tree(0)
    if evaluator == "synth_uniform" and not line:
        u = t["uct"][1:1 + n_root]
        assert (u == u.max()).sum() > 1                    # the last selection at the root still saw an exact tie
    assert sp.stats()["error_flags"] == 0


def test_search_wide_root_lockstep_exact(scamd, orc):
    """> 64 legal moves: the four-round level of the descent (lane owns children lane, lane + 64, ...), the (value, index)
    pair reduction and the header pick of the chosen child, against the oracle with the hash evaluator, with root noise"""
    R = 180
    sp = scamd.SelfPlay(None, n_slots=2, n_games=2, rollout_num=R, num_steps=40, cpuct=2.5, with_noise=True, epsilon=0.15,
                        evaluator="synth", external_noise=True, seed=3)
    st = _pushed(orc, WIDE)
    sp.set_position(1, WIDE)
    srch = orc.Search(st)
    rnd = np.random.RandomState(1)
    wide_levels = 0
    for s in range(R - 1):
        nz = rnd.dirichlet([0.3] * 82).astype(np.float32)
        sp.set_noise(1, nz)
        sp.enqueue(1)
        srch.sim(cpuct=2.5, epsilon=0.15, with_noise=True, noise=nz.astype(np.float64))
        assert list(sp.slot(1)["path"]) == list(srch.last_path()), s
        if s % 9 == 0 or s > R - 4:
            t, d = sp.tree(1), srch.dump()
            assert _same_tree(t, d), s
            wide_levels = int((d["n_child"] > 64).sum())
    assert wide_levels >= 1 and sp.tree(1)["n_child"][0] == 82
    # the chosen children of the wide root come from both rounds (index < 64 and >= 64)
    kids = sp.tree(1)["n"][1:83]
    assert kids[:64].sum() > 0 and kids[64:].sum() > 0
    assert sp.stats()["error_flags"] == 0


@pytest.mark.parametrize("evaluator,oeval", [("synth", "orc_eval_synth"), ("synth_uniform", "orc_eval_synth_uniform"), ("synth_coarse", "orc_eval_synth_coarse")])
def test_search_root_with_more_than_128_children_lockstep_exact(scamd, orc, evaluator, oeval):
    """no search test reached a node with more than 128 children (rounds 3-4 of the wide arg-max ran only
    in the crafted-value test).  Root with 137 legal moves after a 112-ply line (full 8-board history, halfmove clock and
    repetition bookkeeping of a long game): node pools, uct words and paths equal the oracle after every simulation, with the
    hash evaluator and with both tie evaluators (uniform: children are visited last to first, so the first descents take
    children 136, 135, ... from the third round of lanes)."""
    R = 220
    sp = scamd.SelfPlay(None, n_slots=2, n_games=2, rollout_num=R, num_steps=130, cpuct=2.5, with_noise=False, evaluator=evaluator, seed=3)
    st = _pushed(orc, WIDE137)
    assert len(st.legal_moves()) == 137 and st.turn == 1
    sp.set_position(1, WIDE137)
    srch = orc.Search(st)
    for s in range(R - 1):
        sp.enqueue(1)
        srch.sim(evaluator=oeval, cpuct=2.5, with_noise=False)
        path = list(sp.slot(1)["path"])
        assert path == list(srch.last_path()), s
        if evaluator == "synth_uniform" and s == 1:
            assert path[1] == 137, s                  # the LAST maximal child (src/mcts.rs:78-88): node 137 = child 136, from lanes' round 3
                                                      # (later descents follow the oracle: mating moves among the 137 break the pure tie pattern)
        if s % 7 == 0 or s > R - 4:
            assert _same_tree(sp.tree(1), srch.dump()), s
    t = sp.tree(1)
    assert t["n_child"][0] == 137 and np.count_nonzero(t["uct"][1:138]) == 137     # every child's uct word was computed (and compared)
    if evaluator == "synth_uniform":
        assert (t["n"][1 + 128:1 + 137] > 0).all()    # every child of the third round of lanes was chosen (child 128 mates: it then takes the rest)
    assert sp.stats()["error_flags"] == 0


# ---------------------------------------------------------------------------------- network at full depth / width
@pytest.mark.parametrize("nb", [19, 20])
def test_network_matches_reference_goldens_deep(scamd, nb):
    """the reference's default depth (19, py/module.py:110) and BASELINE configs[3] (20 blocks x 256): HIP forward against
    vectors produced by the reference module itself (tools/gen_golden_nn.py)"""
    g = np.load(os.path.join(GOLD, f"nn_ref_b{nb}_c256.npz"))
    eng = scamd.Engine(nb, 256, seed=int(g["seed"]))
    logp, val = eng.forward(g["boards"], g["meta"])
    d = float(np.abs(logp - g["logp"]).max())
    print(f"nb={nb} max|dlogp|={d:.4f} max|dvalue|={float(np.abs(val - g['value']).max()):.5f}")
    np.testing.assert_allclose(logp, g["logp"], rtol=RTOL, atol=3 * ATOL)   # deeper stack: observed max |dlogp| 0.022 (19 blocks), 0.020 (20)
    np.testing.assert_allclose(val, g["value"], rtol=RTOL, atol=ATOL)
    # priors of the legal moves (what the search consumes): total variation distance (validate_inference.py:22-23)
    p_ref, p_hip = np.exp(g["logp"].astype(np.float64)), np.exp(logp.astype(np.float64))
    assert (0.5 * np.abs(p_ref - p_hip).sum(axis=1)).max() < 1e-2
    eng.close()


def test_network_matches_reference_goldens_at_trained_magnitudes(scamd, tmp_path):
    """the reference module's vectors for weights that produce a trained net's output range (tools/gen_golden_nn.py: policy gain x 4,
    value layer x 2; log-probabilities -28..-2, median -13): the bf16 tower within the reference's own rtol = atol = 1e-2
    (scripts/eval_speed.py:40-43; the author saw ~0.08 of bf16 drift on logits of magnitude 12-21, check_model.ipynb cells 6-8),
    priors within 1e-2 total variation (observed: max |dlogp| 0.083 -- the author's 0.08 -- max TV 0.0077).
    The fp8 tower does NOT keep its stated tolerance (prior TV < 0.05) here: that bound was stated and met on
    init-scale outputs (tests/test_gpu_fp8.py); with the policy gain x 4 the same e4m3 feature error is four times the logit error:
    observed max TV 0.145, |dvalue| 0.031.  Recorded with the bound it does meet (0.25): a trained network run in fp8 needs
    calibrated / quantisation-aware weights, which per-channel power-of-two weight scales alone do not give."""
    import scw
    g = np.load(os.path.join(GOLD, "nn_ref_b10_c256_sharp.npz"))
    sd = scw.prng_state_dict(10, 256, int(g["seed"]))
    sd["policy_head.model.3.weight"] = sd["policy_head.model.3.weight"] * np.float32(g["policy_gain_scale"])
    sd["value_head.ffn.2.weight"] = sd["value_head.ffn.2.weight"] * np.float32(g["value_fc2_scale"])
    p = str(tmp_path / "sharp.scw")
    scw.write_scw(p, sd, 10, 256)
    p_ref = np.exp(g["logp"].astype(np.float64))
    for precision in ("bf16", "fp8"):
        eng = scamd.Engine(weights=p, precision=precision)
        logp, val = eng.forward(g["boards"], g["meta"])
        tv = (0.5 * np.abs(p_ref - np.exp(logp.astype(np.float64))).sum(axis=1)).max()
        print(f"{precision}: max|dlogp| {np.abs(logp - g['logp']).max():.4f} (|logp| up to {np.abs(g['logp']).max():.1f}), max|dvalue| "
              f"{np.abs(val - g['value']).max():.5f}, max TV {tv:.5f}")
        if precision == "bf16":
            np.testing.assert_allclose(logp, g["logp"], rtol=RTOL, atol=3 * ATOL)
            np.testing.assert_allclose(val, g["value"], rtol=RTOL, atol=ATOL)
            assert tv < 1e-2
        else:
            assert tv < 0.25 and np.abs(val - g["value"]).max() < 0.05      # NOT the stated 0.05 for the priors: see the docstring
        eng.close()


def test_network_10x128_on_64_positions(scamd, orc):
    """BASELINE configs[1]'s trunk (10 blocks x 128 channels; parity unpinned by the reference, which has no 128-wide
    instantiation) against the oracle's bf16-emulating mode on 64 distinct positions: identical quantisation points,
    only summation order differs.  Bounds = 2x the maxima observed on the MI355X (recorded in DESIGN.md section 7)."""
    games = [g for g in random_games(orc, 90, 120, seed=21) if g[1].legal_moves()][:64]
    boards = np.stack([g[1].encode()[0] for g in games])
    meta = np.stack([g[1].encode()[1] for g in games])
    assert len({b.tobytes() for b in boards}) >= 60
    eng = scamd.Engine(10, 128, seed=9)
    net = orc.Net(10, 128, seed=9, emulate_bf16=True)
    logp, val = eng.forward(boards, meta)
    dl, dv, dp = [], [], []
    for k in range(64):
        ol, ov = net.forward(boards[k], meta[k])
        dl.append(np.abs(logp[k] - ol).max())
        dv.append(abs(val[k] - ov))
        dp.append(0.5 * np.abs(np.exp(logp[k].astype(np.float64)) - np.exp(ol.astype(np.float64))).sum())
    print(f"10x128: max|dlogp|={max(dl):.4f} max|dvalue|={max(dv):.5f} max TVD={max(dp):.5f}")
    assert max(dl) < 4e-2 and max(dv) < 3e-3 and max(dp) < 3.5e-3   # observed 0.0214 / 0.00132 / 0.00171
    eng.close()


# ---------------------------------------------------------------------------------- full size (BASELINE cfg[1], cfg[3])
@pytest.mark.parametrize("nb,C,R", [(10, 128, 180), (10, 256, 180), (20, 256, 800)], ids=["cfg1_10x128_r180", "ref_10x256_r180", "cfg3_20x256_r800"])
def test_full_size_search_invariants(scamd, orc, nb, C, R):
    """256 concurrent games at the BASELINE sizes: invariants that do not need the oracle at size.  configs[1] exactly
    (256 slots, rollout 180, 10x128) and configs[3]'s per-GPU slice (256 slots, rollout 800, 20x256: node pools of
    1 + 800*218 nodes per game, 802 tree positions)"""
    eng = scamd.Engine(nb, C, seed=1)
    G = 256
    sp = scamd.SelfPlay(eng, n_slots=G, n_games=100000, trace_capacity=2 * G, rollout_num=R, num_steps=150, cpuct=2.5, temperature=0.0,
                        temperature_switch=4, epsilon=0.15, with_noise=True, seed=5)
    sp.enqueue(R - 1)
    sp.sync()
    st = sp.stats()
    assert st["error_flags"] == 0 and st["sims_done"] == G * (R - 1) and st["nn_evals"] <= st["sims_done"]
    for g in (0, 97, 255):
        t = sp.tree(g)
        assert t["n"][0] == R - 1 and t["n_child"][0] == 20
        kids = slice(t["first_child"][0], t["first_child"][0] + 20)
        assert t["n"][kids].sum() == R - 2                          # first simulation only expands the root
        assert 0.97 < t["prior"][kids].sum() <= 1.0 + 1e-6           # renormalised by (sum + 1e-5), chess.rs:891
        assert np.isfinite(t["q"]).all() and np.abs(t["q"][0]) <= R and np.isfinite(t["uct"]).all()
        exp = np.nonzero(t["n_child"])[0]
        assert t["n_child"][exp].sum() == len(t["n"]) - 1           # children contiguous, counted once
        # visit counts are consistent down the tree: N(node) = 1 + sum N(children) for every expanded non-root node
        for i in exp[1:][:50]:
            fc, nc = t["first_child"][i], t["n_child"][i]
            assert t["n"][i] == 1 + t["n"][fc:fc + nc].sum()
        # every node's move is legal in its parent's position (replayed with the oracle along the most visited line)
        s, i = orc.State(), 0
        while t["n_child"][i] > 0:
            fc, nc = t["first_child"][i], t["n_child"][i]
            assert [int(m) for m in t["move"][fc:fc + nc]] == s.legal_moves()
            j = fc + int(np.argmax(t["n"][fc:fc + nc]))
            if t["n"][j] == 0:
                break
            s.push(int(t["move"][j]))
            i = j
    sp.enqueue(1)                                                   # the R-th simulation finishes ply 0
    sp.sync()
    assert all(sp.slot(g)["ply"] == 1 for g in (0, 128, 255)) and sp.stats()["plies_done"] == G
    if R == 180:
        sp.enqueue(R)
        sp.sync()
        assert sp.stats()["plies_done"] == 2 * G
    assert sp.stats()["error_flags"] == 0
    sp.close()
    eng.close()


# ---------------------------------------------------------------------------------- --rollout-factor
def test_rollout_factor_games_exact(scamd, orc):
    """-r/--rollout-factor (src/main.rs:175-176): per-ply budget min(300, n_legal * factor), chosen on the device at the
    first simulation of the ply; whole games equal the oracle's"""
    cfg = dict(num_steps=30, cpuct=2.5, temperature=0.0, temperature_switch=4, with_noise=False)
    for factor in (1.5, 0.26, 20.0):
        sp = scamd.SelfPlay(None, n_slots=3, n_games=5, rollout_num=300, rollout_factor=factor, evaluator="synth", seed=17,
                            outcome_gate=100, **cfg)
        sp.run()
        assert sp.stats()["error_flags"] == 0 and sp.stats()["games_finished"] == 5
        for gi in range(5):
            tr = sp.trace(gi)
            ref = orc.selfplay_game(rollout_num=300, rollout_factor=factor, seed=17, game_id=tr["game_id"], outcome_gate=100, **cfg)
            assert tr["steps"] == ref["steps"] and tr["outcome"] == ref["outcome"], (factor, gi)
        budgets = {sum(c[1] for c in s[2]) + 1 for s in sp.trace(0)["steps"]}
        # 20 x n_legal hits the cap of 300 unless the side to move has < 15 legal moves (check evasions)
        assert (max(budgets) == 300) if factor == 20.0 else (len(budgets) > 1 and max(budgets) < 300)
        sp.close()
    with pytest.raises(scamd.EngineError):
        scamd.SelfPlay(None, n_slots=1, rollout_num=100, rollout_factor=2.0, evaluator="synth")


# ---------------------------------------------------------------------------------- trace ring / streaming drain
def _lap(sp, rounds, R):
    """slot 1 burns through game ids (each is set to a mated position and ends after one search) while slot 0 plays on"""
    for _ in range(rounds):
        if sp.slot(1)["status"] == 1:
            sp.set_position(1, MATED)
        sp.enqueue(R)
        sp.sync()


def test_trace_ring_never_hands_a_live_row_to_a_new_game(scamd, orc):
    """ring of 4 rows, 2 slots: slot 0 plays a 12-ply game (row 0) while slot 1 finishes games 1, 2, 3 at once; game 4
    maps to row 0 again and has to WAIT for game 0 (it used to overwrite the live trace).  With trace_hold every trace
    stays until the host has read it: game 0's trace equals the oracle's, bit for bit."""
    R = 8
    cfg = dict(rollout_num=R, num_steps=12, cpuct=2.5, temperature=0.0, temperature_switch=2, with_noise=False)
    sp = scamd.SelfPlay(None, n_slots=2, n_games=9, evaluator="synth", seed=4, outcome_gate=100, trace_capacity=4, trace_hold=True, **cfg)
    _lap(sp, 4, R)
    s1 = sp.slot(1)
    assert s1["status"] == 3 and s1["game_id"] == 4                   # ST_PENDING: waiting for row 0
    got = {}
    first = sp.poll()
    assert sorted(first) == [1, 2, 3]
    for g in first:
        tr = sp.trace(g)
        assert tr["steps"] == [] and tr["outcome"] == {"termination": "Checkmate", "winner": "Black"}
        got[g] = tr
    assert sp.trace(0) is None                                        # still being played
    with pytest.raises(scamd.EngineError):
        sp.run()                                                      # a held ring smaller than n_games needs the poll loop
    for _ in range(200):
        sp.enqueue(R)
        for g in sp.poll():
            assert g not in got
            got[g] = sp.trace(g)
        if len(got) == 9:
            break
    assert sorted(got) == list(range(9)) and sp.stats()["games_finished"] == 9 and sp.stats()["error_flags"] == 0
    for g in (0, 4, 5, 6, 7, 8):
        ref = orc.selfplay_game(seed=4, game_id=g, outcome_gate=100, **cfg)
        assert got[g]["steps"] == ref["steps"] and got[g]["outcome"] == ref["outcome"], g
    sp.poll()                                                         # releases the last batch
    with pytest.raises(scamd.EngineError, match="released|overwritten"):
        sp.trace(0)
    sp.close()
    # without trace_hold a finished trace may be overwritten -- but never a live one: game 4 still waits for game 0,
    # then takes its row; asking for game 0 afterwards is answered with the distinct "overwritten" code
    sp = scamd.SelfPlay(None, n_slots=2, n_games=9, evaluator="synth", seed=4, outcome_gate=100, trace_capacity=4, **cfg)
    _lap(sp, 4, R)
    assert sp.slot(1)["status"] == 3 and sp.trace(0) is None
    sp.run()
    assert sp.stats()["games_finished"] == 9 and sp.stats()["error_flags"] == 0
    with pytest.raises(scamd.EngineError, match="overwritten"):
        sp.trace(0)
    for g in (5, 6, 7, 8):
        tr = sp.trace(g)
        ref = orc.selfplay_game(seed=4, game_id=g, outcome_gate=100, **cfg)
        assert tr["steps"] == ref["steps"] and tr["game_id"] == g
    sp.close()


def test_trace_ring_orders_several_waiting_games_per_row(scamd, orc):
    """ring of 6 rows, 3 slots: slot 0 plays a long game (row 0) while slots 1 and 2 burn through ids; game 6 AND game 12
    both map to row 0 and both wait -- they must take the row one after the other (6 after 0, 12 after 6), not together"""
    R = 8
    cfg = dict(rollout_num=R, num_steps=14, cpuct=2.5, temperature=0.0, temperature_switch=2, with_noise=False)
    sp = scamd.SelfPlay(None, n_slots=3, n_games=15, evaluator="synth", seed=6, outcome_gate=100, trace_capacity=6, trace_hold=True, **cfg)
    got = {}
    both_waited = False
    for it in range(400):
        for slot in (1, 2):
            if it < 12 and sp.slot(slot)["status"] == 1 and sp.slot(slot)["game_id"] not in (6, 12):
                sp.set_position(slot, MATED)             # ids other than 6 and 12 end at once on slots 1 and 2
        sp.enqueue(R)
        st = [sp.slot(k) for k in range(3)]
        both_waited |= sorted(x["game_id"] for x in st if x["status"] == 3) == [6, 12]
        for g in sp.poll():
            assert g not in got
            got[g] = sp.trace(g)
        if len(got) == 15:
            break
    assert both_waited and sorted(got) == list(range(15)) and sp.stats()["error_flags"] == 0
    for g in (0, 6, 12, 13, 14):
        ref = orc.selfplay_game(seed=6, game_id=g, outcome_gate=100, **cfg)
        assert got[g]["steps"] == ref["steps"] and got[g]["outcome"] == ref["outcome"], g
    sp.close()


def test_poll_streams_every_game_once(scamd, orc):
    """sc_selfplay_poll on a bounded ring with many more games than rows: every game is reported exactly once, in time to
    be read, and equals the oracle's game"""
    cfg = dict(rollout_num=6, num_steps=5, cpuct=2.5, temperature=1.0, temperature_switch=100, with_noise=False)
    sp = scamd.SelfPlay(None, n_slots=6, n_games=100, evaluator="synth", seed=9, first_game_id=1000, trace_capacity=12, trace_hold=True, **cfg)
    seen = {}
    for _ in range(400):
        sp.enqueue(6)
        for g in sp.poll(cap=5):                                      # a small cap: the rest is reported by later polls
            assert g not in seen
            seen[g] = sp.trace(g)
        if len(seen) == 100:
            break
    assert sorted(seen) == list(range(100))
    for g in (0, 37, 99):
        ref = orc.selfplay_game(seed=9, game_id=1000 + g, **cfg)
        assert seen[g]["steps"] == ref["steps"] and seen[g]["game_id"] == 1000 + g
    sp.close()


def test_traces_read_while_the_next_ply_runs(scamd, tmp_path):
    """the pipelined streaming loop (lib/sc-selfplay, SelfPlay.stream_traces): finished games are reported by the poll, the next
    ply's steps are enqueued, and only then are the reported traces fetched and written -- a held row is final, so reading it
    beside the running kernels gives byte-identical files to reading everything from an idle handle at the end"""
    eng = scamd.Engine(2, 128, seed=5)
    cfg = dict(n_slots=64, n_games=200, rollout_num=12, num_steps=10, cpuct=2.5, temperature=0.6, temperature_switch=3, with_noise=True, seed=21,
               outcome_gate=0)
    ref = scamd.SelfPlay(eng, **cfg)                       # all trace rows kept, read at the end
    ref.run()
    sp = scamd.SelfPlay(eng, trace_capacity=2 * 64 + 8, trace_hold=True, first_game_id=0, **cfg)
    n = sp.stream_traces(lambda gid: str(tmp_path / f"s{gid}.json"))
    assert n == 200 and sp.stats()["error_flags"] == 0 and sp.stats()["games_finished"] == 200
    for g in range(200):
        ref.write_trace(g, str(tmp_path / f"r{g}.json"))
        assert open(tmp_path / f"r{g}.json").read() == open(tmp_path / f"s{g}.json").read(), g
    sp.close()
    ref.close()
    eng.close()


# ---------------------------------------------------------------------------------- multi-GPU readiness (8e)
def test_disjoint_handles_stand_in_for_ranks(scamd, tmp_path):
    """games shard by id: two handles with disjoint first_game_id ranges -- what two ranks / two GPUs run --
    produce, together, byte-identical trace files to one handle playing all games; `sc-selfplay --gpus <device_count>`
    writes the same files (on the one-GPU box this is the single-device path of the same launcher code)"""
    import subprocess
    cfg = dict(rollout_num=16, num_steps=8, cpuct=2.0, temperature=0.0, temperature_switch=2, epsilon=0.15, with_noise=True, seed=11)
    eng = scamd.Engine(1, 128, seed=11)
    one = scamd.SelfPlay(eng, n_slots=8, n_games=8, **cfg)
    one.run()
    a = scamd.SelfPlay(eng, n_slots=4, n_games=4, first_game_id=0, own_stream=True, **cfg)
    b = scamd.SelfPlay(eng, n_slots=4, n_games=4, first_game_id=4, own_stream=True, **cfg)
    for _ in range(8):
        scamd.enqueue_interleaved([a, b], 16)
    d1, d2, d3 = tmp_path / "one", tmp_path / "two", tmp_path / "cli"
    for d in (d1, d2, d3):
        d.mkdir()
    ids = set()
    for g in range(8):
        one.write_trace(g, str(d1 / f"trace{g + 1}.json"))
    for h in (a, b):
        for g in range(4):
            gid = h.trace(g)["game_id"]
            ids.add(gid)
            h.write_trace(g, str(d2 / f"trace{gid + 1}.json"))
    assert ids == set(range(8))
    ndev = scamd.lib().sc_device_count()
    cli = os.path.join(ROOT, "smart-chess-rust_amd", "lib", "sc-selfplay")
    r = subprocess.run([cli, "-d", "cuda", "--rollout-num", "16", "-n", "8", "--temperature", "0", "--cpuct", "2", "--temperature-switch", "2",
                        "--games", "8", "--concurrency", "8", "--blocks", "1", "--channels", "128", "--seed", "11", "--gpus", str(ndev),
                        "-t", str(d3 / "trace{}.json")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    for k in range(1, 9):
        ref = open(str(d1 / f"trace{k}.json")).read()
        assert open(str(d2 / f"trace{k}.json")).read() == ref, k
        assert open(str(d3 / f"trace{k}.json")).read() == ref, k
    for h in (one, a, b):
        h.close()
    eng.close()


def test_cli_rollout_factor_and_streaming(tmp_path, orc):
    """sc-selfplay -r F: the per-ply budget follows the root's legal-move count; more games than ring rows stream out"""
    import subprocess
    cli = os.path.join(ROOT, "smart-chess-rust_amd", "lib", "sc-selfplay")
    r = subprocess.run([cli, "-d", "cuda", "-r", "1.5", "-n", "4", "--temperature", "0", "--cpuct", "2", "--temperature-switch", "1",
                        "--games", "70", "--concurrency", "2", "--blocks", "1", "--channels", "128", "--seed", "5",
                        "-t", str(tmp_path / "trace{}.json")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    for k in range(1, 71):      # ring = 2*2 + 64 rows < 70 games
        js = json.load(open(str(tmp_path / f"trace{k}.json")))
        st = orc.State()
        assert len(js["steps"]) == 4
        for mv, q, kids in js["steps"]:
            n = len(st.legal_moves())
            assert [c[0] for c in kids] == st.legal_uci() and sum(c[1] for c in kids) == min(300, int(np.float32(n) * np.float32(1.5))) - 1
            st.push(mv)


def test_encode_positions_rows_are_zero_past_n_legal(scamd):
    """the legal-move tables come back zero-filled past n_legal for every batch size (the two tables are separate regions
    of a padded arena)"""
    eng = scamd.Engine(0, 128, seed=1)
    import ctypes as C
    L = scamd.lib()
    for n in (3, 1, 2):
        off = np.zeros(n + 1, np.uint32)
        flat = np.zeros(1, np.uint16)
        lm = np.full((n, 224), 0xFFFF, np.uint16)
        li = np.full((n, 224), 0xFFFF, np.uint16)
        nl = np.zeros(n, np.int32)
        rc = L.sc_encode_positions(eng.h, 0, n, flat.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), None, None,
                                   lm.ctypes.data_as(C.c_void_p), li.ctypes.data_as(C.c_void_p), nl.ctypes.data_as(C.c_void_p), None)
        assert rc == 0 and (nl == 20).all()
        assert (lm[:, 20:] == 0).all() and (li[:, 20:] == 0).all() and (lm[:, :20] != 0).all()
    eng.close()


def test_bench_runs_under_the_launcher_with_rccl(tmp_path):
    """the driver's multi-GPU form of the bench -- torch.distributed.run, one rank per GPU, backend nccl (= RCCL) -- with the
    one rank this box has: rendezvous on 127.0.0.1, barrier, all-reduce of the timing scalars on device tensors are the
    code path of every N (the games themselves never touch a collective)"""
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1",
                        "--no-alt", "--cpu-budget", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e5 and line["error_flags"] == 0 and line["roofline"]["frac"] > 0.1


@pytest.mark.parametrize("C,precision,n_slots", [(128, "bf16", 24), (256, "bf16", 24), (128, "fp8", 24),
                                                 (128, "bf16", 64), (256, "bf16", 128), (128, "fp8", 192), (256, "fp8", 64), (128, "bf16", 256), (128, "fp8", 512)])
def test_fused_step_kernel_equals_the_two_launch_form(scamd, C, precision, n_slots):
    """the fused simulation step (search wave = wave 0 of the tower workgroup, planes handed over in LDS) plays bit-identical
    games to k_mcts + k_tower32 as separate launches (which a handle uses while every launch is timed): moves, visit
    counts, value sums and uct words of every ply, with root noise, temperature sampling and slot recycling.  With whole
    64-slot blocks the step is ONE launch: value_head.ffn.0's tiles are computed by the step kernel's workgroups from feature
    rows handed over inside the launch (uneven load: games end, leaves are terminal, slots idle at the end) -- every partial
    sum feeds a value, every value a backup, so a single stale word shows up in the traces."""
    eng = scamd.Engine(3, C, seed=4, precision=precision)
    n_games = n_slots + n_slots // 2 + 4
    cfg = dict(n_slots=n_slots, n_games=n_games, rollout_num=20, num_steps=9, cpuct=2.5, temperature=0.5, temperature_switch=3, with_noise=True,
               seed=12, outcome_gate=0)
    a = scamd.SelfPlay(eng, **cfg)
    a.run()
    b = scamd.SelfPlay(eng, **cfg)
    b.enable_timing(1)                       # every tower launch bracketed by events: the two-launch form
    b.run()
    assert b.timing(reset=False)["tower_launches"] > 100
    for g in range(n_games):
        ta, tb = a.trace(g), b.trace(g)
        assert ta is not None and ta == tb, g
    assert a.stats() == b.stats() and a.stats()["error_flags"] == 0 and a.stats()["games_finished"] == n_games
    for h in (a, b):
        h.close()
    eng.close()


@pytest.mark.parametrize("where", ["trunk", "policy", "value"])
def test_non_finite_parameters_stay_visible_per_head(scamd, tmp_path, where):
    """a NaN in the network is not swallowed by a ReLU (hardware max maps NaN to 0; the reference warns on non-finite values,
    src/backends/torch.rs:129-135): in the trunk it reaches both outputs, in one head only that head's output -- the heads
    are separate branches (py/module.py:136-152), the other output equals the clean network's bit for bit"""
    import scw
    sd = scw.prng_state_dict(2, 128, 7)
    clean = str(tmp_path / "clean.scw")
    scw.write_scw(clean, sd, 2, 128)
    key = {"trunk": "res_blocks.0.conv1.bias", "policy": "policy_head.model.0.bias", "value": "value_head.conv.0.bias"}[where]
    sd[key] = sd[key].copy()
    sd[key][3] = np.nan
    bad = str(tmp_path / "bad.scw")
    scw.write_scw(bad, sd, 2, 128)
    g = np.load(os.path.join(GOLD, "nn_ref_b1_c256.npz"))
    a, b = scamd.Engine(weights=clean), scamd.Engine(weights=bad)
    la, va = a.forward(g["boards"][:3], g["meta"][:3])
    lb, vb = b.forward(g["boards"][:3], g["meta"][:3])
    assert np.isfinite(la).all() and np.isfinite(va).all()
    if where in ("trunk", "policy"):
        assert np.isnan(lb).all()
    else:
        assert np.array_equal(la, lb)
    if where in ("trunk", "value"):
        assert np.isnan(vb).all()
    else:
        assert np.array_equal(va, vb)
    a.close()
    b.close()


def test_one_launch_form_is_granted_to_one_stream_per_device(scamd):
    """the one-launch step makes workgroups wait for workgroups of the same launch, so two such launches must not run side by
    side on one device: handles of a second engine (its own stream) get the two-launch form while a handle of the first is
    alive, ragged slot counts get it always -- and all of them play the same games"""
    e1, e2 = scamd.Engine(2, 128, seed=9), scamd.Engine(2, 128, seed=9)
    cfg = dict(n_games=80, rollout_num=16, num_steps=8, cpuct=2.5, with_noise=True, seed=3, outcome_gate=0)
    a = scamd.SelfPlay(e1, n_slots=64, **cfg)
    b = scamd.SelfPlay(e2, n_slots=64, **cfg)
    c = scamd.SelfPlay(e1, n_slots=64, **cfg)
    d = scamd.SelfPlay(e1, n_slots=40, **cfg)
    assert [h.launches_per_step() for h in (a, b, c, d)] == [1, 2, 1, 2]
    for _ in range(40):                      # interleaved: the two engines' streams run side by side
        for h in (a, b):
            h.enqueue(8)
    for h in (a, b, c, d):
        h.run()
    for g in range(80):
        t = a.trace(g)
        assert t is not None and t == b.trace(g) == c.trace(g) == d.trace(g), g
    for h in (a, b, c, d):
        assert h.stats()["error_flags"] == 0 and h.stats()["games_finished"] == 80
    a.close(); c.close()
    f = scamd.SelfPlay(e2, n_slots=64, **cfg)     # the first stream's handles are gone: the form is free again
    assert f.launches_per_step() == 1
    for h in (b, d, f):
        h.close()
    e1.close(); e2.close()


def test_in_launch_handoff_wait_is_bounded(scamd):
    """the one-launch step's workgroups wait for each other inside the launch; a wait that can never be satisfied (test aid:
    the arrival target is raised by one) must end by itself: every workgroup gives up after ~0.2 s, the launch completes and
    the handle reports error bit 32 -- nothing hangs.  The failure is LATCHED on the host: the synchronising call
    that first sees the flag and every later enqueue / poll / trace read are refused with SC_ERR_HANDOFF (the values backed up
    since came from stale rows), the statistics stay readable, and handles created afterwards on the device come up in the
    two-launch form, which has no hand-off between workgroups -- and play the right games."""
    import time
    eng = scamd.Engine(1, 128, seed=2)
    cfg = dict(n_slots=64, n_games=64, rollout_num=8, num_steps=4, cpuct=2.5, seed=1)
    ref = scamd.SelfPlay(eng, **cfg)
    ref.enable_timing(1)                # three separate launches: the reference games
    ref.run()
    sp = scamd.SelfPlay(eng, **cfg)
    assert sp.launches_per_step() == 1
    sp.enqueue(4)
    sp.sync()
    assert sp.stats()["error_flags"] == 0
    assert scamd.lib().sc_selfplay_debug_break_handoff(sp.h, 1) == 0
    t0 = time.time()
    sp.enqueue(200)                     # (a whole ply's worth of launches behind the failing one)
    with pytest.raises(scamd.EngineError) as ei:
        sp.sync()
    dt = time.time() - t0
    assert ei.value.code == scamd.binding.ERR_HANDOFF
    assert 0.05 < dt < 3.0, dt          # the first launch waits ~0.2 s (counted on the 100 MHz reference clock); once the flag is up
                                        # later launches give up after 1 ms: a long queue of launches does not stall for minutes
    assert sp.stats()["error_flags"] & 32            # still readable: that is how the host learns what happened
    for call in (lambda: sp.enqueue(1), sp.sync, sp.poll, lambda: sp.trace(0), lambda: sp.run()):
        with pytest.raises(scamd.EngineError) as ei:
            call()
        assert ei.value.code == scamd.binding.ERR_HANDOFF, call
    sp.close()
    # the device is not trusted with the in-launch hand-off again: a fresh handle uses the two-launch step, same games
    sp3 = scamd.SelfPlay(eng, **cfg)
    assert sp3.launches_per_step() == 2
    sp3.run()
    assert sp3.stats()["error_flags"] == 0 and sp3.stats()["games_finished"] == 64
    for g in range(64):
        assert sp3.trace(g) == ref.trace(g), g
    sp3.close()
    ref.close()
    # (test aid) forget the failure: the default form is back for the tests that follow in this process
    assert scamd.lib().sc_debug_clear_handoff_failure(0) == 0 and scamd.lib().sc_debug_clear_handoff_failure(0) == 1
    sp4 = scamd.SelfPlay(eng, **cfg)
    assert sp4.launches_per_step() == 1
    sp4.close()
    # a handle without the one-launch form is not affected by the test aid
    sp2 = scamd.SelfPlay(eng, n_slots=24, n_games=24, rollout_num=8, num_steps=4, cpuct=2.5, seed=1)
    assert scamd.lib().sc_selfplay_debug_break_handoff(sp2.h, 1) == 1
    sp2.run()
    assert sp2.stats()["error_flags"] == 0 and sp2.stats()["games_finished"] == 24
    sp2.close()
    eng.close()


def test_step_form_chosen_for_the_baseline_configurations(scamd):
    """which pipeline form a handle gets is a property of the build (registers and LDS of the fused kernel decide how many of its
    workgroups fit a CU): BASELINE configs[1] (256 bf16 games) and configs[4]'s per-GPU share (512 fp8 games) must get the
    one-launch step -- a kernel change that costs the second fp8 workgroup per CU would otherwise pass every parity test and
    silently lose 10 % (measured once)"""
    bf, f8 = scamd.Engine(1, 128, seed=1), scamd.Engine(1, 128, seed=1, precision="fp8")
    want = [(bf, 256, 1), (bf, 192, 1), (bf, 100, 2), (bf, 512, 3), (f8, 256, 1), (f8, 512, 1), (f8, 500, 2)]
    for eng, slots, launches in want:
        sp = scamd.SelfPlay(eng, n_slots=slots, n_games=slots, rollout_num=4, num_steps=2)
        assert sp.launches_per_step() == launches, (eng.precision, slots, sp.launches_per_step())
        sp.close()
    bf.close()
    f8.close()


@pytest.mark.parametrize("black", [(1, 256, "bf16"), (2, 128, "fp8")])
def test_match_play_one_launch_equals_separate_launches(scamd, black):
    """match play alternates two networks by ply (src/play.rs:318-343): with 64 games the step is one launch whose value FC
    tiles use THIS ply's network while the next launch's value tail finishes the position with the PREVIOUS ply's -- the games
    must equal those of the three-launch form, for players of different width and of different precision"""
    w = scamd.Engine(2, 128, seed=1)
    b = scamd.Engine(black[0], black[1], seed=2, precision=black[2])
    cfg = dict(n_slots=64, n_games=64, rollout_num=10, num_steps=12, cpuct=1.5, temperature=0.0, with_noise=False, outcome_gate=-1, seed=4,
               tie_random=True)
    x = scamd.SelfPlay(w, **cfg)
    x.set_players(w, b)
    assert x.launches_per_step() == 1
    x.run()
    y = scamd.SelfPlay(w, **cfg)
    y.set_players(w, b)
    y.enable_timing(1)
    y.run()
    for g in range(64):
        tx = x.trace(g)
        assert tx is not None and tx == y.trace(g), g
    assert x.stats() == y.stats() and x.stats()["error_flags"] == 0
    for h in (x, y):
        h.close()
    w.close()
    b.close()


@pytest.mark.parametrize("precision,n_slots", [("bf16", 256), ("fp8", 512)])
def test_baseline_configuration_one_launch_equals_separate_launches(scamd, precision, n_slots):
    """BASELINE configs[1] / configs[4] sizing exactly (10 x 128 net, rollout 180, 256 bf16 / 512 fp8 games) for a few plies:
    the one-launch step and the three-launch form play the same games"""
    eng = scamd.Engine(10, 128, seed=1, precision=precision)
    cfg = dict(n_slots=n_slots, n_games=n_slots, rollout_num=180, num_steps=3, cpuct=2.5, temperature=0.0, temperature_switch=4, epsilon=0.15,
               with_noise=True, seed=1234)
    a = scamd.SelfPlay(eng, **cfg)
    assert a.launches_per_step() == 1
    a.run()
    b = scamd.SelfPlay(eng, **cfg)
    b.enable_timing(1)
    b.run()
    for g in range(0, n_slots):
        ta = a.trace(g)
        assert ta is not None and len(ta["steps"]) == 3 and ta == b.trace(g), g
    assert a.stats() == b.stats() and a.stats()["error_flags"] == 0
    for h in (a, b):
        h.close()
    eng.close()
