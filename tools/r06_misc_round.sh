#!/bin/bash
# On the GPU box: round-6 evidence that is not a kernel change -- the GPU fuzz tests, the tiled 4K leg's own counters, the real encoder next to the AVX2 encoder.  usage: tools/r06_misc_round.sh <tag>
T=$1
( timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/${T}_gpu_fuzz.log 2>&1; cat gpurun_out/${T}_gpu_fuzz.log
tools/pmc_leg.sh $T tiles4k > gpurun_out/${T}_pmc_leg_tiles4k.log 2>&1; tail -c 600 gpurun_out/${T}_pmc_leg_tiles4k.log; echo
( timeout 1500 python tools/encoder_fps.py 512 ) > gpurun_out/${T}_encoder_fps.log 2>&1; cat gpurun_out/${T}_encoder_fps.log
