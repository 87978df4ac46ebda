#!/bin/bash
# On the GPU box: instructions of rdoq_block_wave per block (one wavefront per block, RdoqOp) by PMC, for 4x4 .. 32x32 luma blocks and a 4x4 chroma block.
# usage: tools/rdoq_insts.sh <tag>
tag=$1; repo=$PWD
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/${tag}_rdoq_insts.log; : > $out
for cfg in "4 0" "8 0" "16 0" "32 0" "4 2"; do
  set -- $cfg
  d=$repo/gpurun_out/${tag}_rdoq_pmc_$1_$2
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d -- python $repo/tools/rdoq_probe.py $1 $2 > $d.log 2>&1
  grep "^width" $d.log >> $out
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "Rdoq" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print("   ", k[:60], {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
cat $out
