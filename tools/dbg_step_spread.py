"""developer tool: where a step launch's time goes beyond the median workgroup -- start spread, slowest workgroup, per-phase spread
(kernel's own 100 MHz stamps: entry [24], leaf selected [25], network done [26], value-FC tile done [27])"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
eng = scamd.Engine(10, 128, seed=1, precision=os.environ.get("SC_PREC", "bf16"))
G = int(os.environ.get("SC_GAMES", "256"))
sp = scamd.SelfPlay(eng, n_slots=G, n_games=10**6, trace_capacity=2 * G, rollout_num=180, num_steps=150, seed=5)
sp.enqueue(720)
sp.debug_cycles(True)
rows = []
for it in range(30):
    sp.enqueue(5)
    a = sp.debug_cycles(True, read=True).astype(np.int64)[:, 24:28]
    ok = (a[:, 1] > a[:, 0]) & (a[:, 2] > a[:, 1]) & (a[:, 3] >= a[:, 2])
    a = a[ok]
    t0 = a[:, 0].min()
    rows.append(dict(start_spread=(a[:, 0].max() - t0) / 100, extent=(a[:, 3].max() - t0) / 100, med_total=np.median(a[:, 3] - a[:, 0]) / 100,
                     search_med=np.median(a[:, 1] - a[:, 0]) / 100, search_max=(a[:, 1] - a[:, 0]).max() / 100, search_p90=np.percentile(a[:, 1] - a[:, 0], 90) / 100,
                     tower_med=np.median(a[:, 2] - a[:, 1]) / 100, tower_max=(a[:, 2] - a[:, 1]).max() / 100,
                     tile_med=np.median(a[:, 3] - a[:, 2]) / 100, tile_max=(a[:, 3] - a[:, 2]).max() / 100,
                     netdone_spread=(a[:, 2].max() - a[:, 2].min()) / 100, end_spread=(a[:, 3].max() - a[:, 3].min()) / 100))
for k in rows[0]:
    v = np.array([r[k] for r in rows])
    print(f"{k:16s} median {np.median(v):7.2f} us   (min {v.min():7.2f}, max {v.max():7.2f})")
