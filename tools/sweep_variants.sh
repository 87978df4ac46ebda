#!/bin/bash
# On the GPU box: bench every variant built by tools/build_variants.sh (no CPU baselines), one JSON line each.
cd "$(dirname "$0")/.."
for so in kvazaar_amd/lib/variants/*.so; do
  echo "== $so"
  KVZ_HIP_LIB=$PWD/$so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-encoder 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
