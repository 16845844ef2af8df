#!/bin/bash
# developer tool: samples GPU power / clocks (rocm-smi) while a bench configuration runs; prints the samples taken during the run
# usage: tools/power_trace.sh <tag> [bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-pw}; shift
OUT=$R/gpurun_out/$TAG.txt
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $OUT.samples &
SP=$!
python3 $R/bench.py --steps 40 --warmup 2 --repeats 3 --no-alt --cpu-budget 0 "$@" > $OUT.bench.json 2> $OUT.err
kill $SP 2>/dev/null; wait $SP 2>/dev/null
python3 - <<PY
import json
rows=[]
for l in open("$OUT.samples"):
    l=l.strip()
    if not l.startswith("{"): continue
    try: d=json.loads(l)
    except Exception: continue
    c=d.get("card0",{})
    rows.append(c)
keys=[k for k in (rows[0] if rows else {}) if any(s in k.lower() for s in ("power","sclk","mclk","fclk","use"))]
print("samples",len(rows),"keys",keys)
for k in keys:
    vals=[r.get(k) for r in rows]
    print(k, vals[::4])
print(open("$OUT.bench.json").read()[-400:])
PY
