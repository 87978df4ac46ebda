#!/bin/bash
# On the GPU box: the inter CTU pass after a change -- equality with the oracle and launch times on the probe clip (default library), then the lane-0 stage clock of
# the -DKVZ_ICTU_PROFILE variant (kvazaar_amd/lib/variants/libkvz_hip_ictu_prof.so, built ahead).  usage: tools/inter_round.sh <tag> [copies]
tag=$1; copies=${2:-1024}
timeout 300 python tools/inter_ctu_probe.py survey-416x240 $copies > gpurun_out/${tag}_inter_probe.log 2>&1
grep picture gpurun_out/${tag}_inter_probe.log
if [ -f kvazaar_amd/lib/variants/libkvz_hip_ictu_prof.so ]; then
  KVZ_HIP_LIB=$PWD/kvazaar_amd/lib/variants/libkvz_hip_ictu_prof.so timeout 300 python tools/inter_ctu_probe.py survey-416x240 $copies > gpurun_out/${tag}_ictu_stage_profile.log 2>&1
  grep -E "ictu-profile|picture 1" gpurun_out/${tag}_ictu_stage_profile.log | head -20
fi
