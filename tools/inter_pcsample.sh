#!/bin/bash
# On the GPU box: PC-sampling of the inter CTU pass (rocprofv3 beta feature) on tools/inter_ctu_probe.py.  usage: tools/inter_pcsample.sh <tag> [method] [unit] [interval]
tag=$1; method=${2:-host_trap}; unit=${3:-time}; interval=${4:-1}
repo=$PWD
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 -L 2>/dev/null | grep -i -A6 "pc sampl" | head -20
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
  --output-format csv -d $repo/gpurun_out/${tag}_pcs -- python $repo/tools/inter_ctu_probe.py survey-416x240 1024 > $repo/gpurun_out/${tag}_pcs.log 2>&1
echo "rc=$?"
tail -3 $repo/gpurun_out/${tag}_pcs.log
find $repo/gpurun_out/${tag}_pcs -type f | head; du -sh $repo/gpurun_out/${tag}_pcs
cd $repo
python - <<PY
import csv, glob, collections
for p in glob.glob("gpurun_out/${tag}_pcs/**/*pc_sampling*.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    print(p, len(rows), rows[0] if rows else None)
    c = collections.Counter()
    for r in rows:
        key = (r.get("Code_Object_Id", ""), r.get("Code_Object_Offset", r.get("Instruction", "")))
        c[key] += 1
    with open("gpurun_out/${tag}_pcs_hist.csv", "w") as f:
        for (co, off), n in c.most_common():
            f.write("%s,%s,%d\n" % (co, off, n))
    print("distinct", len(c))
PY
ls -la gpurun_out/${tag}_pcs_hist.csv
