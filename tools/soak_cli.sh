#!/bin/bash
# developer tool: CLI soak runs (bf16 256 concurrent / fp8 512 concurrent), every trace file parsed afterwards
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p /tmp/soak1 /tmp/soak2
B=smart-chess-rust_amd/lib/sc-selfplay
t0=$(date +%s)
timeout -k 10 300 $B -d cuda --rollout-num 180 -n 120 -t /tmp/soak1/trace.json --temperature 0 --cpuct 2.5 --temperature-switch 4 --games 1500 --concurrency 256 --blocks 10 --channels 128 --seed 7 > gpurun_out/soak1.log 2>&1 || { echo soak1 failed; tail -5 gpurun_out/soak1.log; exit 1; }
t1=$(date +%s)
timeout -k 10 300 $B -d cuda --rollout-num 100 -n 100 -t /tmp/soak2/trace.json --temperature 0 --cpuct 2.5 --games 1200 --concurrency 512 --blocks 10 --channels 128 --fp8 --seed 9 > gpurun_out/soak2.log 2>&1 || { echo soak2 failed; tail -5 gpurun_out/soak2.log; exit 1; }
t2=$(date +%s)
echo "soak1 $((t1-t0)) s, soak2 $((t2-t1)) s"; tail -2 gpurun_out/soak1.log; tail -2 gpurun_out/soak2.log
python3 - <<PY
import json,glob
for d,n in (("/tmp/soak1",1500),("/tmp/soak2",1200)):
    fs=glob.glob(d+"/trace*.json"); bad=0; plies=0; outc=0
    for f in fs:
        try:
            t=json.load(open(f)); plies+=len(t["steps"]); outc+= t["outcome"] is not None
            assert all(len(s)==3 and len(s[2])>0 for s in t["steps"])
        except Exception as e: bad+=1
    print(d, "files", len(fs), "of", n, "bad", bad, "plies", plies, "with outcome", outc)
PY
