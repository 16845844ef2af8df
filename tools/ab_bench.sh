#!/bin/bash
# developer tool: bench A/B of two builds of the library (smart-chess-rust_amd/lib vs /lib_ab), bf16 256 / fp8 256 / fp8 512 games
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "bf16 256" "fp8 256" "fp8 512"; do set -- $cfg
  for lib in lib lib_ab; do
    SC_ENGINE_LIB=$R/smart-chess-rust_amd/$lib/libsc_engine.so timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-alt --cpu-budget 0 --precision $1 --games $2 > gpurun_out/ab_${lib}_$1_$2.json 2> gpurun_out/ab.err || exit 1
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["repeats"]["values"], d["error_flags"])
PY
