import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
eng = scamd.Engine(10, 256, seed=1)
for (G, R, ng, tc) in [(256, 180, 100000, 512), (256, 180, 1000, 0), (64, 32, 64, 0)]:
    sp = scamd.SelfPlay(eng, n_slots=G, n_games=ng, trace_capacity=tc, rollout_num=R, num_steps=150, cpuct=2.5, seed=5)
    print("created", G, R, ng, tc, sp.stats(), sp.slot(0))
    sp.enqueue(1); print(" after 1", sp.stats(), sp.slot(0))
    sp.enqueue(R - 2); sp.sync(); print(" after R-1", sp.stats(), sp.slot(0))
    sp.close()
