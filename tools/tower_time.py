"""developer tool: average duration of the network tower launch at a given number of positions (HIP events on the launch
stream, every launch timed), from a self-play handle of that many slots.
    [SC_ENGINE_LIB=...] [SC_PREC=fp8] python tools/tower_time.py <channels> <blocks> <slots> [<slots> ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd

C, nb = int(sys.argv[1]), int(sys.argv[2])
for n in [int(x) for x in sys.argv[3:]]:
    eng = scamd.Engine(nb, C, seed=1, precision=os.environ.get("SC_PREC", "bf16"))
    sp = scamd.SelfPlay(eng, n_slots=n, n_games=10 ** 6, trace_capacity=2 * n, rollout_num=180, num_steps=150, seed=5)
    sp.enqueue(40)
    sp.enable_timing(1)
    sp.timing(reset=True)
    steps = int(os.environ.get("SC_TT_STEPS", "200"))   # (long runs: power / clock sampling, tools/power_all.sh)
    sp.enqueue(steps)
    t = sp.timing()
    ms = t["ms_tower_sum"] / max(t["tower_launches"], 1)
    flop = 2.0 * (64 * 112 * 9 * C + nb * (2 * 64 * C * 9 * C + C * C) + 2 * 64 * C * 256 + 64 * 256 * 73)
    print(f"{eng.precision} {nb}x{C} slots {n:5d}: tower {ms * 1e3:8.1f} us  {n / ms / 1e3:7.3f} M positions/s  {n * flop / ms / 1e9:7.1f} TFLOP/s  "
          f"step {t['ms_total'] / steps * 1e3:7.1f} us", flush=True)
    sp.close()
    eng.close()
