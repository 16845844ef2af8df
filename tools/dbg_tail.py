# developer tool (experiment build, tools/build_exp.sh): phases of the fused step kernel's value_head.ffn.0 tail
# usage: SC_ENGINE_LIB=.../lib_exp/libsc_engine.so python tools/dbg_tail.py [bf16|fp8] [slots]
import ctypes as C, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
L = scamd.lib()
L.sc_selfplay_debug_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = scamd.Engine(10, 128, seed=1, precision=prec)
sp = scamd.SelfPlay(eng, n_slots=G, n_games=10000, trace_capacity=2*G, rollout_num=180, num_steps=150, cpuct=2.5, seed=5)
sp.enqueue(400)
L.sc_selfplay_debug_cycles(sp.h, 1, None)
acc = []
for it in range(20):
    sp.enqueue(3)
    eng.L.sc_engine_synchronize(eng.h)
    out = np.zeros((G, 32), np.uint64)
    L.sc_selfplay_debug_cycles(sp.h, 0, out.ctypes.data)
    acc.append(out.astype(np.int64))
a = np.stack(acc)
for n, i, j in [("entry->poll matched", 8, 9), ("acquire+drain", 9, 10), ("stage A", 10, 11), ("mma+store issue", 11, 12), ("store drain", 12, 13), ("whole tail", 8, 13)]:
    x = (a[..., j] - a[..., i]) * 10.0
    print(f"{n:24s} median {np.median(x):8.0f} ns  mean {x.mean():8.0f}  p90 {np.percentile(x, 90):8.0f}  max {x.max():8.0f}")
print("early A loads: fraction of workgroups", a[..., 14].mean())
t0 = a[..., 8]; print("entry spread over the grid (max-min per launch, ns):", np.median((t0.max(1) - t0.min(1)) * 10.0))
t5 = a[..., 13]; print("exit spread:", np.median((t5.max(1) - t5.min(1)) * 10.0), " last exit - last entry:", np.median((t5.max(1) - t0.max(1)) * 10.0))
