#!/bin/bash
# developer tool: a longer CLI soak of the one-launch step (bf16 256 slots, fp8 512 slots): thousands of games through recycled slots,
# every trace file parsed afterwards; the JSON statistics lines go to gpurun_out/soak_long.jsonl
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p /tmp/soakL1 /tmp/soakL2
B=smart-chess-rust_amd/lib/sc-selfplay
timeout -k 10 500 $B -d cuda --rollout-num 180 -n 120 -t /tmp/soakL1/trace.json --temperature 0 --cpuct 2.5 --temperature-switch 4 --games 8000 --concurrency 256 --blocks 10 --channels 128 --seed 17 > gpurun_out/soakL1.log 2>&1 || { echo soakL1 failed; tail -5 gpurun_out/soakL1.log; exit 1; }
echo "bf16 done"
timeout -k 10 400 $B -d cuda --rollout-num 180 -n 120 -t /tmp/soakL2/trace.json --temperature 0 --cpuct 2.5 --temperature-switch 4 --games 8000 --concurrency 512 --blocks 10 --channels 128 --fp8 --seed 19 > gpurun_out/soakL2.log 2>&1 || { echo soakL2 failed; tail -5 gpurun_out/soakL2.log; exit 1; }
grep -h '^{' gpurun_out/soakL1.log gpurun_out/soakL2.log > gpurun_out/soak_long.jsonl; cat gpurun_out/soak_long.jsonl
python3 - <<PY
import json,glob
for d,n in (("/tmp/soakL1",8000),("/tmp/soakL2",8000)):
    fs=glob.glob(d+"/trace*.json"); bad=0; plies=0; outc=0
    for f in fs:
        try:
            t=json.load(open(f)); plies+=len(t["steps"]); outc+= t["outcome"] is not None
            assert all(len(s)==3 and len(s[2])>0 for s in t["steps"])
        except Exception as e: bad+=1
    print(d, "files", len(fs), "of", n, "bad", bad, "plies", plies, "with outcome", outc)
PY
