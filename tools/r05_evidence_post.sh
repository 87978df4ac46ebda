#!/bin/bash
# In the container, after tools/r05_evidence.sh <tag> ran on the GPU box: copies the round's evidence from gpurun_out/ into profiles/ and derives the JSON files bench.py cites.
tag=$1
cp gpurun_out/${tag}_bench_default.json gpurun_out/${tag}_ctu_stage_profile.log gpurun_out/${tag}_smoke.log gpurun_out/${tag}_valu_issue.jsonl profiles/
cp gpurun_out/${tag}_pmc_sq.log profiles/${tag}_pmc_sq.log
cp gpurun_out/${tag}_stats/${tag}_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp gpurun_out/${tag}_pmc_f/${tag}_f_counter_collection.csv profiles/${tag}_f_counter_collection.csv
cp gpurun_out/${tag}_pmc_w/${tag}_w_counter_collection.csv profiles/${tag}_w_counter_collection.csv
python tools/make_pmc_traffic_json.py ${tag} > /dev/null
python tools/valu_mix.py > profiles/${tag}_valu_mix.json
kms=$(python - <<PY
import csv
for r in csv.DictReader(open("profiles/${tag}_kernel_stats.csv")):
    if "intra_ctu_ticket_kernel" in r["Name"]:
        print(float(r["AverageNs"]) / 1e6); break
PY
)
python tools/make_pmc_sq_json.py ${tag} $kms | head -30
ls profiles/${tag}_*
