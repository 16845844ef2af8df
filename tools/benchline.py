"""developer tool: one-line digest of bench.py's JSON line (stdin)"""
import json, sys
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
r = d["roofline"]
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  {r['kernel']} {r['avg_launch_ms']:.4f} ms  frac {r['frac']:.3f}"
      + (f"  2x-games {d['also_2x_games']['value']:.0f}" if d.get("also_2x_games") else "")
      + (f"  alt {d['also']['value']:.0f} tower {d['also']['tower_avg_ms']}" if d.get("also") else ""))
