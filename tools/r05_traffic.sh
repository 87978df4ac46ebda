#!/bin/bash
# On the GPU box: HBM traffic of the CTU kernel from PMC on a 384-frame batch (FETCH_SIZE / WRITE_SIZE, separate passes).  usage: tools/r05_traffic.sh <tag>
tag=$1; repo=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --frames 384 --steps 1 --warmup 1 --no-cpu-baseline --no-ref-encoder --no-extra"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $repo/gpurun_out/${tag}_tf -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $repo/gpurun_out/${tag}_tw -- $B > /dev/null 2>&1
cd $repo
python - <<PY
import csv, glob
for name, d in (("FETCH_SIZE", "tf"), ("WRITE_SIZE", "tw")):
    s = n = 0
    for p in glob.glob(f"gpurun_out/${tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "intra_ctu" in r["Kernel_Name"] and r["Counter_Name"] == name:
                s += float(r["Counter_Value"]); n += 1
    print(f"{name}: {s / n * 1024 / (384 * 510) / 1024:.1f} KB per CTU ({n} launches)")
PY
