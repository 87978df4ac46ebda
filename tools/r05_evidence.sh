#!/bin/bash
# On the GPU box: the evidence round 5 commits under profiles/ for the headline kernel -- the full bench line, rocprofv3 kernel stats of the same command, HBM traffic from
# PMC (FETCH_SIZE / WRITE_SIZE in separate passes), the SQ counters behind `limiter` (tools/pmc_sq.sh), the VALU issue calibration, the stage profile, the GPU suite's log.
# usage: tools/r05_evidence.sh <tag> ; results under gpurun_out/<tag>_*   (then, in the container: tools/r05_evidence_post.sh <tag>)
tag=$1
repo=$PWD
timeout 900 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --no-cpu-baseline --no-ref-encoder --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_stats -o ${tag} -- $B > $repo/gpurun_out/${tag}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $repo/gpurun_out/${tag}_pmc_f -o ${tag}_f -- $B --steps 1 --warmup 1 > $repo/gpurun_out/${tag}_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $repo/gpurun_out/${tag}_pmc_w -o ${tag}_w -- $B --steps 1 --warmup 1 > $repo/gpurun_out/${tag}_pmc_w.log 2>&1
cd $repo
find gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w -name "*.csv" | head -12
timeout 600 tools/pmc_sq.sh ${tag} > gpurun_out/${tag}_pmc_sq.log 2>&1; cat gpurun_out/${tag}_pmc_sq.log
timeout 200 kvazaar_amd/lib/valu_issue_bench > gpurun_out/${tag}_valu_issue.jsonl 2> gpurun_out/${tag}_valu_issue.err; wc -l gpurun_out/${tag}_valu_issue.jsonl
( KVZ_PROFILE_QP=22 timeout 200 python tools/ctu_profile.py 96 ) > gpurun_out/${tag}_ctu_stage_profile.log 2>&1; head -3 gpurun_out/${tag}_ctu_stage_profile.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
