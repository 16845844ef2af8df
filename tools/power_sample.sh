#!/bin/bash
# developer tool: samples socket power and shader clock (rocm-smi, every ~0.1 s) while a command runs on the GPU
# usage: tools/power_sample.sh <tag> <command ...>      -> gpurun_out/<tag>.{samples,out}; prints a digest
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
OUT=$R/gpurun_out/$TAG
( while true; do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | tr -d '\n'; echo; sleep 0.08; done ) > $OUT.samples &
SP=$!
"$@" > $OUT.out 2> $OUT.err
RC=$?
kill $SP 2>/dev/null; wait $SP 2>/dev/null
python3 - <<PY
import json, statistics as st
pw=[]; ck=[]
for l in open("$OUT.samples"):
    l=l.strip()
    if not l.startswith("{"): continue
    try: c=json.loads(l).get("card0",{})
    except Exception: continue
    try:
        p=float(c.get("Current Socket Graphics Package Power (W)","0")); s=int(c.get("sclk clock speed:","(0Mhz)").strip("()").replace("Mhz",""))
    except Exception: continue
    use=int(c.get("GPU use (%)","0") or 0)
    if use>=90 and s>500: pw.append(p); ck.append(s)
if pw: print("$TAG: %d busy samples  power median %.0f W (min %.0f max %.0f)  sclk median %d MHz (min %d max %d)"%(len(pw),st.median(pw),min(pw),max(pw),st.median(ck),min(ck),max(ck)))
else: print("$TAG: no busy samples")
PY
tail -3 $OUT.out | cut -c1-300
exit $RC
