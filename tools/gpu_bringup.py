"""GPU bring-up diagnostics (developer tool; the real parity tests live in tests/ -m gpu).
Runs each stage of the hot path against the CPU oracle and prints where things differ."""
import os
import random
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import oracle_py as orc  # noqa: E402
import scamd  # noqa: E402


def stage(name):
    print(f"\n===== {name}", flush=True)


def random_games(n, maxlen, seed):
    random.seed(seed)
    games = []
    for g in range(n):
        st = orc.State()
        mv = []
        L = random.randint(0, maxlen)
        for _ in range(L):
            lm = st.legal_moves()
            if not lm or st.outcome() is not None and random.random() < 0.3:
                break
            if g % 3 == 0:
                pref = [x for x in lm if abs(st.piece_at(x & 63)) == 2]
                m = random.choice(pref) if pref and random.random() < 0.8 else random.choice(lm)
            else:
                m = random.choice(lm)
            st.push(m)
            mv.append(m)
        games.append((mv, st))
    return games


def check_encode():
    stage("encode_positions vs oracle")
    games = random_games(400, 220, 7)
    t = time.time()
    enc = scamd.encode_positions([g[0] for g in games])
    print("gpu encode time %.3fs for %d positions" % (time.time() - t, len(games)))
    bad = 0
    for i, (mv, st) in enumerate(games):
        ob, om = st.encode()
        ok = np.array_equal(enc["boards"][i], ob) and np.array_equal(enc["meta"][i], om)
        lm = st.legal_moves()
        ok &= list(enc["legal_moves"][i]) == lm
        ok &= list(enc["legal_idx"][i]) == [orc.move_index(m, st.turn) for m in lm]
        oc = st.outcome()
        term = scamd.TERMINATION[int(enc["termination"][i])]
        ok &= (oc["termination"] if oc else None) == term
        ok &= bool(enc["is_check"][i]) == st.is_check()
        ok &= enc["status"][i] == 0
        if not ok:
            bad += 1
            if bad < 5:
                print("MISMATCH game", i, "len", len(mv), st.fen(), "term", term, oc, "meta", enc["meta"][i], om,
                      "boards equal", np.array_equal(enc["boards"][i], ob), "moves equal", list(enc["legal_moves"][i]) == lm)
    print("encode mismatches:", bad, "of", len(games))
    return bad == 0


def check_nn():
    stage("network forward vs reference goldens")
    ok_all = True
    for nb in (1, 10):
        g = np.load(os.path.join(ROOT, "tests", "golden", f"nn_ref_b{nb}_c256.npz"))
        eng = scamd.Engine(nb, 256, seed=int(g["seed"]))
        logp, val = eng.forward(g["boards"], g["meta"])
        dl = np.abs(logp - g["logp"]).max(axis=1)
        dv = np.abs(val - g["value"])
        print(f"blocks={nb}: max|dlogp| per position {np.round(dl, 4)}  |dv| {np.round(dv, 4)}")
        print("   sum exp(logp):", np.exp(logp.astype(np.float64)).sum(axis=1)[:4])
        tol = 1e-2 + 1e-2 * np.abs(g["logp"])
        ok = bool((np.abs(logp - g["logp"]) <= tol).all() and (dv <= 1e-2 + 1e-2 * np.abs(g["value"])).all())
        print("   within reference tolerance (rtol=atol=1e-2):", ok)
        ok_all &= ok
        # bf16-emulating oracle: much tighter
        net = orc.Net(nb, 256, seed=int(g["seed"]), emulate_bf16=True)
        k = 0
        ol, ov, lat = net.forward(g["boards"][k], g["meta"][k], latent=True)
        print("   vs bf16-emulating oracle: max|dlogp| %.2e dv %.2e" % (np.abs(logp[k] - ol).max(), abs(val[k] - ov)))
        if not ok or np.abs(logp[k] - ol).max() > 5e-2:
            for stg in [0] + list(range(1, nb + 1)):
                d = eng.debug(g["boards"][k:k + 1], g["meta"][k:k + 1], stg)[0]
                print("      stage", stg, "gpu mean %.4f std %.4f finite %s" % (d.mean(), d.std(), np.isfinite(d).all()))
            d = eng.debug(g["boards"][k:k + 1], g["meta"][k:k + 1], 1000)[0]
            print("      latent max|gpu-oracle| %.3e (oracle std %.3f)" % (np.abs(d - lat).max(), lat.std()))
        eng.close()
    # C=128 variant vs the oracle (no reference instantiation exists)
    for C in (128,):
        eng = scamd.Engine(2, C, seed=5)
        net = orc.Net(2, C, seed=5, emulate_bf16=True)
        g = np.load(os.path.join(ROOT, "tests", "golden", "nn_ref_b1_c256.npz"))
        logp, val = eng.forward(g["boards"][:3], g["meta"][:3])
        for k in range(3):
            ol, ov = net.forward(g["boards"][k], g["meta"][k])
            print(f"C={C} pos {k}: max|dlogp| %.2e dv %.2e" % (np.abs(logp[k] - ol).max(), abs(val[k] - ov)))
        eng.close()
    return ok_all


def check_predict():
    stage("predict (gather/exp/renorm) vs oracle")
    g = np.load(os.path.join(ROOT, "tests", "golden", "nn_ref_b1_c256.npz"))
    eng = scamd.Engine(1, 256, seed=int(g["seed"]))
    hip = scamd.ChessHip(eng)
    steps, pri, val = hip.predict(["e2e4", "e7e5"])
    st = orc.State()
    st.push("e2e4")
    st.push("e7e5")
    lm = st.legal_moves()
    idx = [orc.move_index(m, st.turn) for m in lm]
    b, m = st.encode()
    ref = np.exp(g["logp"][0] * 0)  # placeholder
    net = orc.Net(1, 256, seed=int(g["seed"]))
    ol, ov = net.forward(b, m)
    e = np.exp(ol[idx])
    ref = e / (e.sum() + 1e-5)
    print("steps equal:", steps == lm, " prior TVD %.4e" % (0.5 * np.abs(pri - ref).sum()), " dv %.3e" % abs(val - ov))
    eng.close()
    return steps == lm


def check_search_synth():
    stage("search with the synthetic evaluator vs oracle (exact)")
    R = 64
    sp = scamd.SelfPlay(None, n_slots=4, n_games=4, rollout_num=R, num_steps=30, cpuct=2.5, temperature=0.0,
                        temperature_switch=4, with_noise=False, evaluator="synth", seed=3)
    st = orc.State()
    srch = orc.Search(st)
    bad = 0
    for s in range(R - 1):
        sp.enqueue(1)
        sp.sync()
        srch.sim(cpuct=2.5, with_noise=False)
        t = sp.tree(0)
        d = srch.dump()
        same = (len(t["n"]) == len(d["n"]) and np.array_equal(t["n"], d["n"]) and np.array_equal(t["q"], d["q"])
                and np.array_equal(t["uct"], d["uct"]) and np.array_equal(t["move"][1:], d["move"][1:]))
        if not same:
            bad += 1
            if bad < 4:
                print("sim", s, "nodes", len(t["n"]), len(d["n"]), "path gpu", sp.slot(0)["path"], "oracle", srch.last_path())
                k = min(len(t["n"]), len(d["n"]))
                w = np.nonzero((t["n"][:k] != d["n"][:k]) | (t["q"][:k] != d["q"][:k]) | (t["uct"][:k] != d["uct"][:k]))[0][:5]
                print("  first diffs at", w, t["n"][w], d["n"][w], t["q"][w], d["q"][w], t["uct"][w], d["uct"][w])
    print("mismatching simulations:", bad, "of", R - 1, " stats", sp.stats())
    sp.close()
    # whole games, temperature sampling in the first plies, vs oracle self-play
    sp = scamd.SelfPlay(None, n_slots=8, n_games=8, rollout_num=24, num_steps=150, cpuct=2.5, temperature=0.0,
                        temperature_switch=4, with_noise=False, evaluator="synth", seed=11)
    t = time.time()
    sp.run()
    print("8 synth games: %.2fs" % (time.time() - t), sp.stats())
    gbad = 0
    for gi in range(8):
        tr = sp.trace(gi)
        ref = orc.selfplay_game(rollout_num=24, num_steps=150, cpuct=2.5, temperature=0.0, temperature_switch=4,
                                with_noise=False, seed=11, game_id=tr["game_id"])
        same = tr["steps"] == ref["steps"] and tr["outcome"] == ref["outcome"]
        if not same:
            gbad += 1
            print("game", gi, "len", len(tr["steps"]), len(ref["steps"]), tr["outcome"], ref["outcome"])
            for i, (a, b) in enumerate(zip(tr["steps"], ref["steps"])):
                if a != b:
                    print("  first differing ply", i, a[0], b[0], a[1], b[1])
                    break
    print("mismatching games:", gbad, "of 8")
    sp.close()
    return bad == 0 and gbad == 0


def check_selfplay_net():
    stage("self-play with the network (small) + timing")
    eng = scamd.Engine(10, 256, seed=1)
    for G in (64, 256):
        sp = scamd.SelfPlay(eng, n_slots=G, n_games=100000, rollout_num=180, num_steps=150, cpuct=2.5, seed=5)
        sp.enable_timing(1)
        sp.enqueue(20)
        sp.sync()
        sp.timing(reset=True)
        t = time.time()
        sp.enqueue(360)
        sp.sync()
        dt = time.time() - t
        tm = sp.timing()
        st = sp.stats()
        print(f"G={G}: 360 sim steps in {dt:.3f}s -> {G * 360 / dt:.0f} sims/s; tower avg "
              f"{tm['ms_tower_sum'] / max(tm['tower_launches'], 1):.3f} ms, span {tm['ms_total']:.1f} ms, stats {st}")
        sp.close()
    eng.close()
    eng = scamd.Engine(10, 128, seed=1)
    sp = scamd.SelfPlay(eng, n_slots=256, n_games=100000, rollout_num=180, num_steps=150, cpuct=2.5, seed=5)
    sp.enable_timing(1)
    sp.enqueue(20)
    sp.sync()
    sp.timing(reset=True)
    t = time.time()
    sp.enqueue(360)
    sp.sync()
    dt = time.time() - t
    tm = sp.timing()
    print(f"C=128 G=256: {256 * 360 / dt:.0f} sims/s; tower avg {tm['ms_tower_sum'] / max(tm['tower_launches'], 1):.3f} ms", sp.stats())
    sp.close()
    eng.close()
    return True


if __name__ == "__main__":
    which = sys.argv[1:] or ["encode", "nn", "predict", "synth", "net"]
    fns = {"encode": check_encode, "nn": check_nn, "predict": check_predict, "synth": check_search_synth,
           "net": check_selfplay_net}
    res = {}
    for w in which:
        try:
            res[w] = fns[w]()
        except Exception:
            traceback.print_exc()
            res[w] = False
    print("\nSUMMARY", res)
