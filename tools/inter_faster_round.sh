#!/bin/bash
# On the GPU box: the inter CTU pass at `--preset faster` (quarter-sample search, CABAC coefficient cost at every QP) -- GPU tests, the real encoder's bitstream,
# the pass timed on 1024 sequences with rocprofv3 kernel stats, and bench.py's config 4 leg on its own.  usage: tools/inter_faster_round.sh <tag>
tag=$1
repo=$PWD
timeout 240 python -m pytest tests/test_gpu_inter_ctu.py tests/test_entropy_inter.py -m gpu -q -k "faster or rejects" > gpurun_out/${tag}_faster_tests.log 2>&1
tail -4 gpurun_out/${tag}_faster_tests.log
timeout 120 python -m pytest tests/test_e2e_dropin.py -m gpu -q -k "test_inter_pass_inside_the_encoder and (faster or qp32 or veryfast-416x240)" > gpurun_out/${tag}_faster_e2e.log 2>&1
tail -4 gpurun_out/${tag}_faster_e2e.log
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_faster_stats -o ${tag}_faster -- python $repo/tools/inter_ctu_probe.py faster-owf-qp27 1024 > $repo/gpurun_out/${tag}_faster_probe.log 2>&1
cd $repo
grep "picture" gpurun_out/${tag}_faster_probe.log
timeout 200 python tools/inter_leg_probe.py 192 > gpurun_out/${tag}_inter_leg.json 2> gpurun_out/${tag}_inter_leg.err
tail -30 gpurun_out/${tag}_inter_leg.json
