#!/bin/bash
# On the GPU box: PC-sampling histogram of the CTU kernel (rocprofv3 beta feature).  usage: tools/pcsample.sh <method> <unit> <interval>
repo=$PWD
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 \
  --output-format csv -d $repo/gpurun_out/pcs_$1 -- python $repo/bench.py --frames 96 --steps 2 --warmup 1 --no-cpu-baseline --no-ref-encoder > $repo/gpurun_out/pcs_$1.log 2>&1
echo "rc=$?"
tail -5 $repo/gpurun_out/pcs_$1.log
find $repo/gpurun_out/pcs_$1 -type f | head; du -sh $repo/gpurun_out/pcs_$1
