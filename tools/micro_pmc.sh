#!/bin/bash
# On the GPU box: HBM traffic of every streaming kernel of bench_kernels.py from PMC counters (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes,
# --kernel-trace only), next to its duration.  usage: tools/micro_pmc.sh <tag> ; result gpurun_out/<tag>_micro_pmc.json
tag=$1
repo=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $repo/gpurun_out/${tag}_mpmc_$c -o m -- python $repo/bench_kernels.py --reps 3 --warmup 1 > $repo/gpurun_out/${tag}_mpmc_$c.log 2>&1
done
cd $repo
python - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter); dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob("gpurun_out/${tag}_mpmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(p)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvz::", "")
            if name.startswith("__amd") or not name.startswith("dev_"): continue
            key = (name, int(r["Grid_Size"]))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
            if c == "FETCH_SIZE": dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
for key in sorted(acc):
    f = acc[key]["FETCH_SIZE"] / max(1, cnt[key]["FETCH_SIZE"]) * 1024; w = acc[key]["WRITE_SIZE"] / max(1, cnt[key]["WRITE_SIZE"]) * 1024
    d = sorted(dur[key])[len(dur[key]) // 2] if dur[key] else 0
    out.append({"kernel": key[0], "grid": key[1], "launches": cnt[key]["FETCH_SIZE"], "fetch_bytes": f, "write_bytes": w, "median_ns": d,
                "traffic_GBps": round((f + w) / d, 1) if d else None})
json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench_kernels.py --reps 3 --warmup 1; counters in KB x 1024; "
           "MI355X_MICROARCH.md: FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950 -- raw values reported", "kernels": out}, open("gpurun_out/${tag}_micro_pmc.json", "w"), indent=1)
for o in out: print(o)
PY
