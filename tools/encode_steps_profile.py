"""developer tool: the trace -> training-tensor path (sc_encode_steps) on 256 quick self-play games, for
`rocprofv3 --kernel-trace --stats -- python3 tools/encode_steps_profile.py` (per-kernel time of k_encode_positions / k_steps_dist)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
eng = scamd.Engine(1, 128, seed=1)
quick = scamd.SelfPlay(eng, n_slots=256, n_games=256, rollout_num=8, num_steps=int(os.environ.get("SC_PLIES", "100")), cpuct=2.5, temperature=0.0,
                       temperature_switch=8, epsilon=0.15, with_noise=True, seed=5, outcome_gate=10 ** 6)
quick.run()
games = []
for g in range(256):
    tr = quick.trace(g)
    games.append([(s[0], [(c[0], c[1]) for c in s[2]]) for s in tr["steps"]])
quick.close()
for it in range(4):
    t0 = time.perf_counter()
    r = scamd.encode_steps_batch(games, engine=eng)
    wall = time.perf_counter() - t0
    k_ms, call_ms = scamd.binding.encode_steps_last_timing()
    P = int(r["ply_off"][-1])
    print(f"{P} plies: kernels {k_ms:.3f} ms ({P / k_ms / 1e3:.2f} M plies/s, {P * 26336 / k_ms / 1e6:.1f} GB/s written), call {call_ms:.1f} ms, python {wall * 1e3:.0f} ms", flush=True)
eng.close()
