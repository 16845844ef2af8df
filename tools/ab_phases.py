"""developer tool: the step launch's phase split (in-kernel wall-clock stamps of the stamped instantiation) for the library named by
SC_ENGINE_LIB -- to see WHERE a build gained or lost:  SC_ENGINE_LIB=.../libsc_engine.so python tools/ab_phases.py [games] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
import bench
games = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
eng = scamd.Engine(10, 128, seed=1, precision=prec)
sp = scamd.SelfPlay(eng, n_slots=games, n_games=10 ** 7, rollout_num=180, num_steps=150, cpuct=2.5, temperature=0.0, temperature_switch=4, epsilon=0.15,
                    with_noise=True, seed=1234, outcome_gate=100, trace_capacity=4 * games)
sp.enqueue(720)
sp.sync()
print(os.environ.get("SC_ENGINE_LIB", "lib"), {k: v for k, v in bench.step_phases(sp, 180).items() if k != "note"}, flush=True)
