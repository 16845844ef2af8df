"""developer tool: latency of the Level-1 `Game::predict` path (one position per call from the host)"""
import sys, time
sys.path.insert(0, "smart-chess-rust_amd")
import numpy as np, scamd
line = ["e2e4", "c7c5", "g1f3", "d7d6", "d2d4", "c5d4", "f3d4", "g8f6", "b1c3"]
for C in (128, 256):
    eng = scamd.Engine(10, C, seed=1)
    hip = scamd.ChessHip(eng)
    for _ in range(5): hip.predict(line)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): enc = scamd.encode_positions([line], engine=eng)
    t1 = time.perf_counter()
    for _ in range(n): eng.predict(enc["boards"], enc["meta"], [enc["legal_idx"][0]])
    t2 = time.perf_counter()
    for _ in range(n): hip.predict(line)
    t3 = time.perf_counter()
    print(f"C={C}: encode_positions {1e3*(t1-t0)/n:.3f} ms, predict_batch(1) {1e3*(t2-t1)/n:.3f} ms, ChessHip.predict {1e3*(t3-t2)/n:.3f} ms per call")
    eng.close()
