#!/bin/bash
# On the GPU box: SQ / traffic counters of the inter CTU pass (rocprofv3 --pmc passes, no trace domains), on tools/inter_ctu_probe.py's launch of `copies` sequences.
# usage: tools/inter_pmc.sh <tag> [case] [copies]
tag=$1; name=${2:-survey-416x240}; copies=${3:-1024}
repo=$PWD
cd /tmp && export TMPDIR=/tmp
P="python $repo/tools/inter_ctu_probe.py $name $copies"
run() { timeout 300 rocprofv3 --pmc "${@:2}" --output-format csv -d $repo/gpurun_out/${tag}_pmc_$1 -- $P > $repo/gpurun_out/${tag}_pmc_$1.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
run c SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM
run f FETCH_SIZE
run w WRITE_SIZE
cd $repo
python - <<PY
import csv, glob, collections, json
out = {}
for p in sorted(glob.glob("gpurun_out/${tag}_pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        if "inter_ctu" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc:
        out[k] = {"per_launch": acc[k] / n[k], "launches": n[k]}
        print(f"{k:24s} {acc[k]/n[k]:.5g} per launch ({n[k]} launches)")
json.dump({"kernel": "inter_ctu_ticket_kernel", "workload": "tools/inter_ctu_probe.py $name $copies", "counters": out}, open("gpurun_out/${tag}_inter_pmc.json", "w"), indent=1)
PY
