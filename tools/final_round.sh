#!/bin/bash
# On the GPU box: the evidence a round commits under profiles/ when the CTU kernel changed late -- GPU suite, smoke, the full bench line, the preset /
# QP variants without their CPU legs, the stage profiles (needs kvazaar_amd/lib/variants/libkvz_hip_prof.so built ahead) and the rocprofv3 kernel stats
# of the bench command.  usage: tools/final_round.sh <tag> ; results under gpurun_out/<tag>_*
T=${1:-final}
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/${T}_pytest_gpu.log; cat gpurun_out/${T}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; cut -c1-300 gpurun_out/${T}_bench.json
for P in faster fast medium-pu13; do timeout 100 python bench.py --preset $P --qp 27 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_${P}_qp27.json 2>/dev/null; done
timeout 100 python bench.py --preset fast --qp 22 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_fast_qp22.json 2>/dev/null
timeout 100 python bench.py --qp 32 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_qp32.json 2>/dev/null
timeout 100 python bench.py --qp 37 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_qp37.json 2>/dev/null
for f in gpurun_out/${T}_bench_*.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['verified'])"; done
if [ -z "$NO_STAGE_PROFILE" ]; then
( KVZ_PROFILE_S32=1 KVZ_PROFILE_CABAC=1 KVZ_PROFILE_QP=22 timeout 120 python tools/ctu_profile.py 32; KVZ_PROFILE_QP=32 timeout 120 python tools/ctu_profile.py 32; KVZ_PROFILE_QP=22 timeout 120 python tools/ctu_profile.py 32 ) > gpurun_out/${T}_ctu_stage_profile.log 2>&1
fi
if [ -z "$NO_ROCPROF" ]; then
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -o ${T} -- python $R/bench.py --no-cpu-baseline --no-ref-encoder --no-extra > $R/gpurun_out/${T}_stats.log 2>&1
find $R/gpurun_out/${T}_stats -name "*kernel_stats.csv" | head -2
fi
