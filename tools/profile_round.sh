#!/bin/bash
# On the GPU box: the measurements a round commits under profiles/ -- full bench line, rocprofv3 kernel stats of the same
# command, and HBM traffic of the CTU kernel from PMC (FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains
# besides --kernel-trace).  usage: tools/profile_round.sh <tag> ; results under gpurun_out/<tag>_*
tag=$1
repo=$PWD
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --no-cpu-baseline --no-ref-encoder --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_stats -o ${tag} -- $B > $repo/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $repo/gpurun_out/${tag}_pmc_f -o ${tag}_f -- $B --steps 1 --warmup 1 > $repo/gpurun_out/${tag}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $repo/gpurun_out/${tag}_pmc_w -o ${tag}_w -- $B --steps 1 --warmup 1 > $repo/gpurun_out/${tag}_pmc_w.log 2>&1
cd $repo
cat gpurun_out/${tag}_bench.json
find gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w -name "*.csv" | head -20
# per-kernel view: bench_kernels.py lines and the rocprofv3 kernel stats of the same command
python bench_kernels.py > gpurun_out/${tag}_micro_kernels.jsonl 2> gpurun_out/${tag}_micro.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_micro_stats -o ${tag}_micro -- python $repo/bench_kernels.py --reps 5 > $repo/gpurun_out/${tag}_micro_stats.log 2>&1
cd $repo
# the inter CTU pass (BASELINE config 4): rocprofv3 kernel stats of tools/inter_ctu_probe.py (1024 sequences of the 416x240 survey clip, every picture checked against the oracle)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_inter_stats -o ${tag}_inter -- python $repo/tools/inter_ctu_probe.py survey-416x240 1024 > $repo/gpurun_out/${tag}_inter_probe.log 2>&1
cd $repo
tail -8 gpurun_out/${tag}_inter_probe.log
# the entropy coder (kvz_hip_batch_entropy_code) on the headline batch: rocprofv3 kernel stats of tools/entropy_probe.py, and the real encoder's frame rates (AVX2 / device
# search / device search + device entropy coding) from tools/encoder_fps.py
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/${tag}_entropy_stats -o ${tag}_entropy -- python $repo/tools/entropy_probe.py 1536 22 > $repo/gpurun_out/${tag}_entropy_probe.log 2>&1
cd $repo
tail -3 gpurun_out/${tag}_entropy_probe.log
python tools/encoder_fps.py 512 ultrafast 22 > gpurun_out/${tag}_encoder_fps.log 2>&1
tail -3 gpurun_out/${tag}_encoder_fps.log
