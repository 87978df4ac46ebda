#!/bin/bash
# On the GPU box: the inter CTU pass after a change -- equality with the oracle and launch times on the probe clip (default library), then the lane-0 stage clock of
# the -DKVZ_ICTU_PROFILE variant (kvazaar_amd/lib/variants/libkvz_hip_ictu_prof.so, built ahead), then any other variants named.  usage: tools/inter_round.sh <tag> [copies = 4096: enough sequences to keep every workgroup slot busy] [variant...]
tag=$1; copies=${2:-4096}; shift; shift
export KVZ_HIP_INTER_VERBOSE=1
timeout 300 python tools/inter_ctu_probe.py survey-416x240 $copies > gpurun_out/${tag}_inter_probe.log 2>&1
grep -E "picture|workgroups per CU" gpurun_out/${tag}_inter_probe.log | sort | uniq -c | sort -rn | head -8
if [ -f kvazaar_amd/lib/variants/libkvz_hip_ictu_prof.so ]; then
  KVZ_HIP_LIB=$PWD/kvazaar_amd/lib/variants/libkvz_hip_ictu_prof.so timeout 300 python tools/inter_ctu_probe.py survey-416x240 $copies > gpurun_out/${tag}_ictu_stage_profile.log 2>&1
  grep -E "ictu-profile|picture 1" gpurun_out/${tag}_ictu_stage_profile.log | head -20
fi
for v in "$@"; do
  KVZ_HIP_LIB=$PWD/kvazaar_amd/lib/variants/libkvz_hip_$v.so timeout 200 python tools/inter_ctu_probe.py survey-416x240 $copies 2>&1 | grep -E "picture|workgroups per CU" | sort | uniq -c | sort -rn | head -4 | sed "s/^/$v: /" | tee -a gpurun_out/${tag}_variants.log
done
