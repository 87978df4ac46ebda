#!/bin/bash
# On the GPU box: SQ issue/wait counters for the CTU kernel (two rocprofv3 --pmc passes, no trace domains).
# usage: tools/pmc_sq.sh <tag> [bench args...]
tag=$1; shift
repo=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ref-encoder --no-extra $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  --output-format csv -d $repo/gpurun_out/pmc_${tag}_a -- $B > $repo/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  --output-format csv -d $repo/gpurun_out/pmc_${tag}_b -- $B > $repo/gpurun_out/pmc_${tag}_b.log 2>&1
# lane utilisation: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64 lanes x 4 [quad-cycles]) = active lanes per issued VALU instruction
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA \
  --output-format csv -d $repo/gpurun_out/pmc_${tag}_c -- $B > $repo/gpurun_out/pmc_${tag}_c.log 2>&1
cd $repo
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/pmc_${tag}_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        if "intra_ctu" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f"{k:24s} {acc[k]/n[k]:.4g} per launch ({n[k]} launches)")
PY
