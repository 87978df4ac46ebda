#!/bin/bash
# On the GPU box: the medium (RDOQ) CTU kernel after a change -- its GPU parity tests, the C3 bench leg (96 pictures), the stage profile under load.  usage: tools/r06_medium_round.sh <tag> [notests]
T=$1
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
( timeout 900 python -m pytest tests/test_rdoq.py tests/test_encoder_parity.py tests/test_gpu_ctu.py -m gpu -x -q -k "rdoq or medium or nxn" 2>&1 | tail -5 ) > gpurun_out/${T}_medium_tests.log 2>&1; cat gpurun_out/${T}_medium_tests.log
fi
timeout 600 python bench.py --only medium --medium-frames 96 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_only_medium.json; python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_only_medium.json')); print('medium: %.0f CTUs/s kernel %.1f ms verified %s' % (d['value'], d['kernel_ms'], d['verified']))"
( KVZ_PROFILE_RDOQ=1 KVZ_PROFILE_NXN=1 KVZ_PROFILE_QP=22 timeout 200 python tools/ctu_profile.py 224 ) > gpurun_out/${T}_prof_medium.log 2>&1; cat gpurun_out/${T}_prof_medium.log
