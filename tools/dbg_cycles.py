import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
L = scamd.lib()
L.sc_selfplay_debug_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
eng = scamd.Engine(10, 128, seed=1)
G = 256
R = int(os.environ.get("SC_DBG_ROLLOUT", "180"))
sp = scamd.SelfPlay(eng, n_slots=G, n_games=10000, trace_capacity=512, rollout_num=R, num_steps=150, cpuct=2.5, seed=5)
sp.enqueue(400)
L.sc_selfplay_debug_cycles(sp.h, 1, None)
acc = []
for it in range(20):
    sp.enqueue(3)
    eng.L.sc_engine_synchronize(eng.h)   # plain stream sync: no flush launch, stamps are from the last full k_mcts
    out = np.zeros((G, 32), np.uint64)
    L.sc_selfplay_debug_cycles(sp.h, 0, out.ctypes.data)
    acc.append(out.astype(np.int64))
a = np.stack(acc)  # [it][G][8]
names = ["expand (0->1)", "ctl+descent (2->3)", "leafpos+rep (3->4)", "stage+movegen (4->5)", "idx+encode (5->6)", "whole kernel (0->6)"]
d = [a[..., 1] - a[..., 0], a[..., 3] - a[..., 2], a[..., 4] - a[..., 3], a[..., 5] - a[..., 4], a[..., 6] - a[..., 5], a[..., 6] - a[..., 0]]
for n, x in zip(names, d):
    x = x[(x > 0) & (x < 10**7)]
    if x.size == 0:
        print(f"{n:24s} (no stamps)")
        continue
    print(f"{n:24s} median {np.median(x):9.0f}  mean {x.mean():9.0f}  p90 {np.percentile(x, 90):9.0f}  max {x.max():9.0f}  (s_memtime ticks)")
dep = a[..., 7].astype(np.float64)
des = (a[..., 3] - a[..., 2]).astype(np.float64)
ok = (des > 0) & (des < 10**7)
print(f"levels walked: median {np.median(dep[ok]):.1f} mean {dep[ok].mean():.2f} max {dep[ok].max():.0f}")
A = np.stack([dep[ok], np.ones(ok.sum())], 1)
coef = np.linalg.lstsq(A, des[ok], rcond=None)[0]
print(f"descent ticks ~= {coef[0]:.0f} per level + {coef[1]:.0f}")
