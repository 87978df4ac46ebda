#!/bin/bash
# On the GPU box: the quick loop of round 5 -- CTU-pass parity tests, the headline line without its auxiliary legs, SQ instruction counters of the same kernel on a
# smaller batch, the stage profile.  usage: tools/r05_quick.sh <tag> [tests...]   results under gpurun_out/<tag>_*
tag=$1; shift
repo=$PWD
tests=${*:-tests/test_gpu_ctu.py tests/test_encoder_parity.py}
timeout 900 python -m pytest $tests -x -q -m gpu > gpurun_out/${tag}_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${tag}_gputest.log
timeout 300 python bench.py --no-extra --no-cpu-baseline --no-ref-encoder > gpurun_out/${tag}_bench_quick.json 2> gpurun_out/${tag}_bench_quick.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_quick.json"))
print("CTUs/s", d["value"], "ms/step", d["ms_per_step"], "verified", d["verified"])
PY
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --frames 384 --steps 2 --warmup 1 --no-cpu-baseline --no-ref-encoder --no-extra"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
  --output-format csv -d $repo/gpurun_out/${tag}_pmc_a -- $B > $repo/gpurun_out/${tag}_pmc_a.log 2>&1
cd $repo
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/${tag}_pmc_a/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        if "intra_ctu" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    ctus = 384 * 510
    for k in acc: print(f"{k:24s} {acc[k]/n[k]:.5g} per launch, {acc[k]/n[k]/ctus:.5g} per CTU ({n[k]} launches)")
PY
KVZ_PROFILE_QP=22 timeout 300 python tools/ctu_profile.py 96 > gpurun_out/${tag}_ctu_stage_profile.log 2>&1; cat gpurun_out/${tag}_ctu_stage_profile.log
