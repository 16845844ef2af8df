#!/bin/bash
# PMC passes on the tower kernel (developer tool). usage: tools/pmc.sh <tag> <channels>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc}; CH=${2:-256}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-alt --cpu-budget 0 --channels $CH"
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_tower" in k or "k_select" in k or "k_value" in k or "k_expand" in k:
            a=acc[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/summary.txt","w") as o:
    for k,d in acc.items():
        o.write(k+"\n")
        for c,(s,n) in sorted(d.items()): o.write("   %-32s avg/launch %.4g  (n=%d)\n"%(c,s/n,n))
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*.csv" -size +1M -delete; rm -f $OUT/counters.txt.big
