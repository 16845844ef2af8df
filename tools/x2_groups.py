"""developer tool: 512 bf16 games per GPU as two interleaved groups (two HIP streams), in the launch form the handles pick
(experiment builds: SC_FUSED=1 forces the fused step kernel + value FC launch for own-stream handles too)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smart-chess-rust_amd"))
import scamd
eng = scamd.Engine(10, 128, seed=1, precision=os.environ.get("SC_PREC", "bf16"))
K, G = 2, int(os.environ.get("SC_GAMES", "512"))
sps = [scamd.SelfPlay(eng, n_slots=G // K, n_games=10**6, trace_capacity=2 * G, rollout_num=180, num_steps=150, seed=5, first_game_id=k * 10**6, own_stream=True) for k in range(K)]
print("launches per step", [sp.launches_per_step() for sp in sps], flush=True)
scamd.enqueue_interleaved(sps, 360)
for sp in sps: sp.sync()
s0 = sum(sp.stats()["sims_done"] for sp in sps); t0 = time.perf_counter()
scamd.enqueue_interleaved(sps, 1800)
for sp in sps: sp.sync()
dt = time.perf_counter() - t0
s1 = sum(sp.stats()["sims_done"] for sp in sps)
print(f"{G} games in {K} groups: {(s1 - s0) / dt / 1e6:.3f} M sims/s, {dt / 1800 * 1e6:.1f} us per step pair, err {[sp.stats()['error_flags'] for sp in sps]}", flush=True)
