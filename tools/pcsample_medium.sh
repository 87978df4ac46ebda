#!/bin/bash
# On the GPU box: PC-sampling histogram of the medium (RDOQ) CTU kernel (rocprofv3 beta feature).  usage: tools/pcsample_medium.sh <tag> <method> <unit> <interval>
repo=$PWD; tag=$1
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $2 --pc-sampling-unit $3 --pc-sampling-interval $4 \
  --output-format csv -d $repo/gpurun_out/${tag}_pcs -- python $repo/bench.py --only medium --medium-frames 96 --no-cpu-baseline --no-ref-encoder > $repo/gpurun_out/${tag}_pcs.log 2>&1
echo "rc=$?"
tail -3 $repo/gpurun_out/${tag}_pcs.log
find $repo/gpurun_out/${tag}_pcs -type f | head; du -sh $repo/gpurun_out/${tag}_pcs
f=$(find $repo/gpurun_out/${tag}_pcs -name "*pc_sampling*.csv" | head -1)
[ -n "$f" ] && head -5 $f && wc -l $f
