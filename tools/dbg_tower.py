import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
g = np.load(os.path.join(ROOT, "tests", "golden", "nn_ref_b10_c256.npz"))
boards = np.repeat(g["boards"], 32, axis=0)[:256]
meta = np.repeat(g["meta"], 32, axis=0)[:256]
names = ["conv1", "LN1+store", "conv2", "LN2(+SE w loads)", "pool+SE", "resid+store", "heads"]
for C in (128, 256):
    eng = scamd.Engine(10, C, seed=1)
    for _ in range(3):
        d = eng.debug(boards, meta, 2000)
    t = d[:, 0, :8].astype(np.float64)   # [pos][8]
    med = np.median(t, axis=0)
    tot = med[:7].sum()
    print(f"C={C}: total stamped {tot:.0f} cycles  " + "  ".join(f"{n}={v:.0f} ({100*v/tot:.0f}%)" for n, v in zip(names, med)))
    eng.close()
