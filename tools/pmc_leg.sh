#!/bin/bash
# On the GPU box: the counters of one auxiliary leg of bench.py (inter | medium | intra4k | tiles4k | entropy) -> gpurun_out/<tag>_pmc_leg_<leg>.json (copy to profiles/): three
# rocprofv3 --pmc passes (SQ issue / wait, FETCH_SIZE, WRITE_SIZE; no trace domains) of `python bench.py --only <leg>`, then a --kernel-trace --stats pass for the durations.
# usage: tools/pmc_leg.sh <tag> <leg>
tag=$1; leg=$2
repo=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --only $leg --no-cpu-baseline --no-ref-encoder"
run() { timeout 400 rocprofv3 --pmc "${@:2}" --output-format csv -d $repo/gpurun_out/${tag}_${leg}_pmc_$1 -- $B > $repo/gpurun_out/${tag}_${leg}_pmc_$1.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run f FETCH_SIZE
run w WRITE_SIZE
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_${leg}_stats -o ${tag}_${leg} -- $B > $repo/gpurun_out/${tag}_${leg}_stats.log 2>&1
cd $repo
python tools/make_pmc_leg_json.py $tag $leg
