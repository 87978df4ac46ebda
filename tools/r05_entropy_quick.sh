#!/bin/bash
# On the GPU box: the entropy coder's parity tests and its leg at two batch sizes with the host phase clock.  usage: tools/r05_entropy_quick.sh <tag>
tag=$1
timeout 900 python -m pytest tests/test_gpu_entropy.py tests/test_entropy_inter.py -x -q -m gpu > gpurun_out/${tag}_gputest_entropy.log 2>&1; echo "pytest entropy rc=$?"; tail -3 gpurun_out/${tag}_gputest_entropy.log
for n in 768 1536; do
  KVZ_HIP_ENTROPY_TIMES=1 timeout 300 python bench.py --only entropy --entropy-pictures $n 2> gpurun_out/${tag}_ent_${n}.err | tee gpurun_out/${tag}_ent_${n}.json | cut -c1-200
  grep "kvz_hip entropy" gpurun_out/${tag}_ent_${n}.err | tail -4
done
