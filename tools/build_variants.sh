#!/bin/bash
# Builds libkvz_hip.so variants of the CTU kernel (threads per CTU x waves per SIMD) for tools/sweep_variants.sh.
set -e
cd "$(dirname "$0")/.."
mkdir -p kvazaar_amd/lib/variants
for v in ${VARIANTS:-256:3 256:4 512:6 512:8}; do
  t=${v%%:*}; w=${v##*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -DKVZ_CTU_THREADS=$t -DKVZ_CTU_WAVES_PER_EU=$w $EXTRA -Rpass-analysis=kernel-resource-usage \
    -o kvazaar_amd/lib/variants/libkvz_hip_t${t}_w${w}.so kvazaar_amd/csrc/kvz_hip.hip 2>&1 \
    | grep -A12 "intra_ctu_ticket" | grep -E "VGPRs:|Scratch|Occupancy \[|LDS" | sed "s/.*remark: */t$t w$w: /" &
done
wait
