#!/bin/bash
# On the GPU box: kernels and memory copies of chain_full with uploads, on one timeline.  usage: tools/chain_h2d_trace.sh <tag>
repo=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $repo/gpurun_out/$1_h2d_trace -o t -- python $repo/tools/chain_h2d_ab.py 1536 1 ${2:-2} > $repo/gpurun_out/$1_h2d_trace.log 2>&1
cd $repo; tail -1 gpurun_out/$1_h2d_trace.log
python - gpurun_out/$1_h2d_trace <<'PY'
import csv, glob, sys
d = sys.argv[1]
ev = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:50])))
for p in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s" % r.get("Direction", r.get("Kind", "?"))))
ev.sort()
t0 = ev[0][0]
big = [e for e in ev if e[1] - e[0] > 3e5]
for a, b, n in big[-44:]:
    print("%9.1f %9.1f %8.1f ms  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, n))
PY
