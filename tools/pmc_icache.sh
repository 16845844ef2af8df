#!/bin/bash
# developer tool: instruction-cache and scalar-cache counters of the bench's kernels (one PMC pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 1 --warmup 1 --repeats 1 --no-alt --cpu-budget 0 "$@" > $OUT/p1.out 2> $OUT/p1.err || echo "pass failed"
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/p1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        a=acc[r["Kernel_Name"][:70]][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/summary.txt","w") as o:
    for k,d in acc.items():
        if "k_" not in k: continue
        o.write(k+"\n")
        for c,(s,n) in sorted(d.items()): o.write("   %-32s avg/launch %.6g  (n=%d)\n"%(c,s/n,n))
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*.csv" -size +1M -delete
