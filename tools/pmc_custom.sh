#!/bin/bash
# On the GPU box: one rocprofv3 --pmc pass over the given counters for the CTU kernel.  usage: tools/pmc_custom.sh <tag> "<counters>" [bench args]
tag=$1; ctrs=$2; shift 2
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $repo/gpurun_out/pmcx_${tag} -- python $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ref-encoder "$@" > $repo/gpurun_out/pmcx_${tag}.log 2>&1
cd $repo
grep -i "error\|invalid\|not " gpurun_out/pmcx_${tag}.log | head -5
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/pmcx_${tag}/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        if "intra_ctu" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f"{k:28s} {acc[k]/n[k]:.4g} per launch ({n[k]} launches)")
PY
