import sys, os, time
sys.path.insert(0, "smart-chess-rust_amd")
import scamd
eng = scamd.Engine(10, 128, seed=1)
for n in (256, 512):
    sp = scamd.SelfPlay(eng, n_slots=n, n_games=10**6, trace_capacity=2*n, rollout_num=180, num_steps=150, seed=5)
    print(n, "launches per step", sp.launches_per_step(), flush=True)
    sp.enqueue(360); sp.sync()
    s0 = sp.stats()["sims_done"]; t0 = time.perf_counter()
    sp.enqueue(1800); sp.sync()
    dt = time.perf_counter() - t0
    st = sp.stats()
    print(f"  {n} games: {(st['sims_done'] - s0) / dt / 1e6:.3f} M sims/s, {dt / 1800 * 1e6:.1f} us per step, err {st['error_flags']}", flush=True)
    sp.close()
eng.close()
