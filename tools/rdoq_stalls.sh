#!/bin/bash
# On the GPU box: where one wavefront's cycles go in rdoq_block_wave (RdoqOp, one block): wait / issue counters of the SQ.  usage: tools/rdoq_stalls.sh <tag> <width>
tag=$1; w=$2; repo=$PWD
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/${tag}_rdoq_stalls_$w.log; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); d=$repo/gpurun_out/${tag}_rdoq_st_${w}_$i
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $d -- python $repo/tools/rdoq_probe.py $w 0 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { grep -i "error\|invalid\|not" $d.log | head -3 >> $out; continue; }
  python - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "Rdoq" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print({c: round(v / n[c]) for c, v in acc.items()})
PY
done
cat $out
