#!/bin/bash
# On the GPU box: the entropy coder's two forms of stage 3 side by side (byte-at-a-time / 32-bit units, lanes per wavefront), its parity tests, and the headline's
# quick loop.  How profiles/r05_ea_* were made, at commit 85c7276 where both forms were in the library (KVZ_HIP_ENTROPY_CODER=bytes selected the old one; it has
# been removed since: the variable is ignored now and every line of this script measures the 32-bit form).  usage: tools/r05_entropy_ab.sh <tag>   results under gpurun_out/<tag>_*
tag=$1
timeout 900 python -m pytest tests/test_gpu_entropy.py tests/test_entropy_inter.py -x -q -m gpu > gpurun_out/${tag}_gputest_entropy.log 2>&1; echo "pytest entropy rc=$?"; tail -3 gpurun_out/${tag}_gputest_entropy.log
for n in 768 1536; do
  for cfg in "bytes 16" "wide 64" "wide 32" "wide 16"; do
    set -- $cfg
    echo "== pictures $n coder $1 lanes $2"
    KVZ_HIP_ENTROPY_TIMES=1 KVZ_HIP_ENTROPY_CODER=$1 KVZ_HIP_ENTROPY_LANES=$2 timeout 300 python bench.py --only entropy --entropy-pictures $n 2> gpurun_out/${tag}_ent_${1}_${2}_${n}.err | tee gpurun_out/${tag}_ent_${1}_${2}_${n}.json | cut -c1-200
    tail -12 gpurun_out/${tag}_ent_${1}_${2}_${n}.err | grep "kvz_hip entropy" | tail -6
  done
done
timeout 900 python -m pytest tests/test_gpu_ctu.py -x -q -m gpu > gpurun_out/${tag}_gputest.log 2>&1; echo "pytest ctu rc=$?"; tail -2 gpurun_out/${tag}_gputest.log
timeout 300 python bench.py --no-extra --no-cpu-baseline --no-ref-encoder > gpurun_out/${tag}_bench_quick.json 2> gpurun_out/${tag}_bench_quick.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_quick.json"))
print("CTUs/s", d["value"], "ms/step", d["ms_per_step"], "verified", d["verified"])
PY
