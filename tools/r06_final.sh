#!/bin/bash
# On the GPU box: everything round 6 commits under profiles/ for the final state of the code (one box, one run): the GPU suite, the headline evidence (tools/r05_evidence.sh:
# bench line, rocprofv3 kernel stats of the same command, PMC traffic, SQ counters, stage profile, smoke), the other QPs / presets, the streaming kernels, the counters of
# every auxiliary leg incl. the tiled one, the medium kernel's stage profile under load and the RDOQ routine's instruction counts, the real encoder.  usage: tools/r06_final.sh <tag>
T=$1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/${T}_gputest.log; cat gpurun_out/${T}_gputest.log
tools/r05_evidence.sh $T
for P in faster fast medium-pu13; do timeout 100 python bench.py --preset $P --qp 27 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_${P}_qp27.json 2>/dev/null; done
timeout 100 python bench.py --preset fast --qp 22 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_fast_qp22.json 2>/dev/null
timeout 100 python bench.py --qp 32 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_qp32.json 2>/dev/null
timeout 100 python bench.py --qp 37 --no-extra --no-ref-encoder --no-cpu-baseline > gpurun_out/${T}_bench_qp37.json 2>/dev/null
timeout 300 python bench.py --width 3840 --height 2160 --tiles 4x2 --frames 384 --no-extra --no-cpu-baseline --no-ref-encoder > gpurun_out/${T}_bench_4k_tiles4x2_f384.json 2>/dev/null
timeout 600 python bench.py --preset veryfast-inter --tiles 4x2 --frames 400 --steps 3 --warmup 1 > gpurun_out/${T}_bench_tiles4x2_veryfast_inter_400.json 2>/dev/null
for f in gpurun_out/${T}_bench_*.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['verified'])"; done
timeout 400 python bench_kernels.py > gpurun_out/${T}_micro_kernels.jsonl 2> gpurun_out/${T}_micro.err; wc -l gpurun_out/${T}_micro_kernels.jsonl
for leg in intra4k tiles4k medium inter entropy; do timeout 900 tools/pmc_leg.sh $T $leg > gpurun_out/${T}_pmc_leg_${leg}.log 2>&1; tail -c 300 gpurun_out/${T}_pmc_leg_${leg}.log; echo; done
( KVZ_PROFILE_RDOQ=1 KVZ_PROFILE_NXN=1 KVZ_PROFILE_QP=22 timeout 200 python tools/ctu_profile.py 224 ) > gpurun_out/${T}_prof_medium.log 2>&1; head -2 gpurun_out/${T}_prof_medium.log
tools/rdoq_insts.sh $T 2>&1 | grep -v simple_timer | tail -10
( timeout 1500 python tools/encoder_fps.py 512 ) > gpurun_out/${T}_encoder_fps.log 2>&1; cat gpurun_out/${T}_encoder_fps.log
