"""developer tool: per-wave phase stamps of the tower (needs the experiment build: tools/build_exp.sh, then
SC_ENGINE_LIB=smart-chess-rust_amd/lib_exp/libsc_engine.so python tools/dbg_tower.py)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
g = np.load(os.path.join(ROOT, "tests", "golden", "nn_ref_b10_c256.npz"))
N = int(os.environ.get("SC_DBG_N", "256"))          # positions per launch (512 with an occupancy-2 build: two workgroups per CU)
boards = np.repeat(g["boards"], (N + 7) // 8, axis=0)[:N]
meta = np.repeat(g["meta"], (N + 7) // 8, axis=0)[:N]
names = ["conv1", "LN1 stats", "LN1 barrier", "LN1 norm", "store+barrier", "conv2", "LN2 stats", "LN2 barrier", "LN2 norm",
         "pool", "pool barrier", "fc1+fc2", "resid+stores", "end barrier", "prologue+stem", "gather+end", "value head", "policy conv1+LN", "policy conv2+LN+scatter", "softmax", "v: conv", "v: bias+stats", "v: LN barrier", "v: LN apply", "p1: conv", "p1: stats", "p1: LN barrier", "p1: apply+store", "p2: conv", "p2: stats", "p2: LN barrier", "p2: apply", "p2: barrier", "stem: prologue", "stem: conv", "stem: stats", "stem: LN barrier", "stem: apply", "prologue", "-"]
for C in [int(x) for x in os.environ.get("SC_DBG_C", "128,256").split(",")]:
    eng = scamd.Engine(10, C, seed=1, precision=os.environ.get("SC_PREC", "bf16"))
    for _ in range(3):
        d = eng.debug(boards, meta, 2000)
    t = d.reshape(d.shape[0], -1)[:, :160].astype(np.float64).reshape(-1, 4, 40)[:, :, :40]   # [pos][wave][16]
    med = np.median(t, axis=0)                                  # [wave][16]
    tot = med[0].sum()
    print(f"C={C}: wave-0 total {tot:.0f} cycles", flush=True)
    for k, n in enumerate(names):
        print(f"   {n:14s} " + "  ".join(f"{med[w, k]:8.0f}" for w in range(4)) + f"   ({100 * med[:, k].mean() / tot:.1f}%)")
    eng.close()
