#!/bin/bash
# On the GPU box: the round's closing evidence for the inter CTU pass after its interpolation / SATD rework -- the pass timed with rocprofv3 kernel stats (QP 22 probe,
# `faster` probe), bench.py's config 4 leg, then the whole GPU suite and smoke().  usage: tools/inter_final_round.sh <tag>
tag=$1
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_inter_stats -o ${tag}_inter -- python $repo/tools/inter_ctu_probe.py survey-416x240 1024 > $repo/gpurun_out/${tag}_inter_probe.log 2>&1
cd $repo
grep picture gpurun_out/${tag}_inter_probe.log
timeout 100 python tools/inter_ctu_probe.py faster-owf-qp27 1024 > gpurun_out/${tag}_faster_probe.log 2>&1
grep picture gpurun_out/${tag}_faster_probe.log
timeout 100 python tools/inter_ctu_probe.py cabac-coeff-cost-qp32 1024 > gpurun_out/${tag}_inter_probe_qp32.log 2>&1
grep picture gpurun_out/${tag}_inter_probe_qp32.log
timeout 200 python tools/inter_leg_probe.py 192 > gpurun_out/${tag}_inter_leg.json 2> gpurun_out/${tag}_inter_leg.err
grep -E '"value"|"verified"|pass_ms' -A0 gpurun_out/${tag}_inter_leg.json | head
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/${tag}_gputest.log 2>&1
tail -4 gpurun_out/${tag}_gputest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1
tail -2 gpurun_out/${tag}_smoke.log
