#!/bin/bash
# On the GPU box: the probe clip through the default library and the variants named on the command line (kvazaar_amd/lib/variants/libkvz_hip_<name>.so).  usage: tools/inter_variants.sh <tag> <name>...
tag=$1; shift
timeout 200 python tools/inter_ctu_probe.py survey-416x240 1024 2>&1 | grep picture | sed "s/^/default: /" | tee gpurun_out/${tag}_variants.log
for v in "$@"; do
  KVZ_HIP_LIB=$PWD/kvazaar_amd/lib/variants/libkvz_hip_$v.so timeout 200 python tools/inter_ctu_probe.py survey-416x240 1024 2>&1 | grep picture | sed "s/^/$v: /" | tee -a gpurun_out/${tag}_variants.log
done
