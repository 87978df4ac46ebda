#!/bin/bash
# On the GPU box: what the round commits for the inter CTU pass's CABAC coefficient pricing (picture QP >= 28) -- the new GPU tests first, then the pass timed at
# QP 22 (regression check of the fast-cost mode) and at QP 32 with rocprofv3 kernel stats, then the whole GPU suite.  usage: tools/inter_cabac_round.sh <tag>
tag=$1
repo=$PWD
timeout 300 python -m pytest tests/test_gpu_inter_ctu.py tests/test_entropy_inter.py -m gpu -q -k "not 2160p and not 1080p" > gpurun_out/${tag}_inter_tests.log 2>&1
tail -4 gpurun_out/${tag}_inter_tests.log
timeout 200 python -m pytest tests/test_e2e_dropin.py -m gpu -q -k "qp32 or qp24 or veryfast-416x240" > gpurun_out/${tag}_e2e_tests.log 2>&1
tail -4 gpurun_out/${tag}_e2e_tests.log
timeout 120 python tools/inter_ctu_probe.py survey-416x240 1024 > gpurun_out/${tag}_inter_probe_qp22.log 2>&1
tail -7 gpurun_out/${tag}_inter_probe_qp22.log
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_inter_qp32_stats -o ${tag}_inter_qp32 -- python $repo/tools/inter_ctu_probe.py cabac-coeff-cost-qp32 1024 > $repo/gpurun_out/${tag}_inter_probe_qp32.log 2>&1
cd $repo
grep "picture" gpurun_out/${tag}_inter_probe_qp32.log
timeout 120 python tools/inter_ctu_probe.py fast-pan-owf-qp37 1024 > gpurun_out/${tag}_inter_probe_qp37.log 2>&1
grep "picture" gpurun_out/${tag}_inter_probe_qp37.log
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/${tag}_gputest.log 2>&1
tail -4 gpurun_out/${tag}_gputest.log
