# developer tool (experiment build, tools/build_exp.sh): where the fused step kernel's expansion spends its cycles, from the kernel's
# first instruction to the end of dev_expand.   SC_ENGINE_LIB=.../lib_exp/libsc_engine.so python tools/dbg_expand.py
import ctypes as C, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
L = scamd.lib()
L.sc_selfplay_debug_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
eng = scamd.Engine(10, 128, seed=1)
G = 256
sp = scamd.SelfPlay(eng, n_slots=G, n_games=10000, trace_capacity=512, rollout_num=180, num_steps=150, cpuct=2.5, seed=5)
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
for n, i, j in [("kernel entry -> search starts", 21, 0), ("entry -> loads issued", 0, 16), ("loads issued -> ctl in SGPRs", 16, 17), ("ctl -> value ready", 17, 18), ("value -> children written", 18, 19), ("backward", 19, 20), ("rest of expand", 20, 1), ("whole expand", 0, 1)]:
    x = a[..., j] - a[..., i]; x = x[(x > 0) & (x < 10**6)]
    if x.size == 0:
        print(f"{n:32s} (no stamps: needs an experiment build, tools/build_exp.sh)")
        continue
    print(f"{n:32s} median {np.median(x):8.0f}  mean {x.mean():8.0f}  p90 {np.percentile(x, 90):8.0f}")
