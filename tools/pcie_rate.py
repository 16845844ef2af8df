import sys, time
sys.path.insert(0, "smart-chess-rust_amd")
import numpy as np, scamd
g = np.load("tests/golden/nn_ref_b10_c256.npz")
boards = np.repeat(g["boards"], 32, axis=0)[:256]
meta = np.repeat(g["meta"], 32, axis=0)[:256]
enc = scamd.encode_positions([[]])
li = [enc["legal_idx"][0]] * 256
for C in (128, 256):
    eng = scamd.Engine(10, C, seed=1)
    for _ in range(3): eng.predict(boards, meta, li)
    t0 = time.perf_counter(); n = 30
    for _ in range(n): eng.predict(boards, meta, li)
    dt = (time.perf_counter() - t0) / n
    print(f"C={C}: sc_predict_batch 256 positions from host buffers: {dt*1e3:.3f} ms per call = {256/dt:.0f} positions/s (PCIe-inclusive)")
    eng.close()
