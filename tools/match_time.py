import sys, time
sys.path.insert(0, "smart-chess-rust_amd")
import scamd
a, b = scamd.Engine(10, 128, seed=1), scamd.Engine(10, 128, seed=2)
t0 = time.perf_counter()
r = scamd.play_match(a, b, n_games=100, rollout=100, cpuct=1.5, temperature=0.0, temperature_switch=0, num_steps=200, seed=3, swap=True)
print("both colour assignments:", round(time.perf_counter() - t0, 3), "s", r["as_white"]["results"], r["as_black"]["results"])
t0 = time.perf_counter()
r1 = scamd.play_match(a, b, n_games=100, rollout=100, cpuct=1.5, temperature=0.0, temperature_switch=0, num_steps=200, seed=3, swap=False)
print("one assignment:", round(time.perf_counter() - t0, 3), "s", r1["as_white"]["results"], r1["as_white"]["traces"] == r["as_white"]["traces"])
