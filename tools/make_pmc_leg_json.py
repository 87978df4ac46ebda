#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc_leg_<leg>.json from the passes of tools/pmc_leg.sh: every counter of the leg's kernel(s) summed over the kernels of ONE invocation of the leg and
divided by the leg's units (CTUs or pictures; `units_per_launch` of the JSON `bench.py --only <leg>` printed), plus the kernels' average durations.
usage: tools/make_pmc_leg_json.py <tag> <leg>"""
import collections, csv, glob, json, sys
tag, leg = sys.argv[1], sys.argv[2]
KERNELS = {"inter": ["inter_ctu_ticket_kernel"], "medium": ["intra_ctu_ticket_kernel"], "intra4k": ["intra_ctu_ticket_kernel"], "tiles4k": ["intra_ctu_ticket_kernel"], "entropy": ["dev_entropy_"]}[leg]
units, dps = None, 1  # dps: dispatches of the leg's kernel per invocation (the tiled leg launches one batch per tile size side by side)
for line in open(f"gpurun_out/{tag}_{leg}_pmc_a.log"):
    line = line.strip()
    if line.startswith("{"):
        try:
            units = json.loads(line).get("units_per_launch")
            dps = json.loads(line).get("dispatches_per_step", 1)
        except ValueError:
            pass
per_unit, launches = {}, {}
for p in sorted(glob.glob(f"gpurun_out/{tag}_{leg}_pmc_*/**/*counter_collection.csv", recursive=True)):
    acc, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        if any(k in r["Kernel_Name"] for k in KERNELS):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add((r["Kernel_Name"], r["Dispatch_Id"]))
    names = sorted({k for k, _ in next(iter(disp.values()), set())})
    for c in acc:
        # invocations of the leg inside one process = dispatches of the first kernel name
        n_inv = max(1, len([1 for k, _ in disp[c] if k == names[0]]) // dps) if names else 1
        per_unit[c] = acc[c] / n_inv / units if units else None
        launches[c] = n_inv
durations = {}
for p in glob.glob(f"gpurun_out/{tag}_{leg}_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if any(k in r["Name"] for k in KERNELS):
            durations[r["Name"][:100]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
out = {"leg": leg, "kernels": KERNELS, "units_per_launch": units, "unit": "picture" if leg == "entropy" else "CTU", "per_unit": per_unit, "invocations_in_pass": launches,
       "kernel_stats": durations,
       "method": "rocprofv3 --pmc, four passes (tools/pmc_leg.sh), of `python bench.py --only %s`; FETCH_SIZE / WRITE_SIZE in KB, raw (MI355X_MICROARCH.md: uncalibrated for narrow accesses)" % leg}
json.dump(out, open(f"gpurun_out/{tag}_pmc_leg_{leg}.json", "w"), indent=1)
print(json.dumps(out)[:1500])
