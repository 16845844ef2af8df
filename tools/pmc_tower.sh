#!/bin/bash
# developer tool: PMC passes on the tower kernel alone (tools/tower_time.py workload).
# usage: tools/pmc_tower.sh <tag> <channels> <blocks> <slots> [lib]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc}; CH=${2:-128}; NB=${3:-10}; SLOTS=${4:-256}
[ -n "$5" ] && export SC_ENGINE_LIB=$5
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_IFETCH" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/tower_time.py $CH $NB $SLOTS > $OUT/p$i.out 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_tower" in k:
            a=acc[k[:60]][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/summary.txt","w") as o:
    for k,d in acc.items():
        o.write(k+"\n")
        for c,(s,n) in sorted(d.items()): o.write("   %-32s avg/launch %.5g  (n=%d)\n"%(c,s/n,n))
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/p1.out
