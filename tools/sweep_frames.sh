#!/bin/bash
# On the GPU box: throughput vs batch size for the given library variants.
cd "$(dirname "$0")/.."
for so in "$@"; do
  for f in 96 192 384; do
    echo -n "$so frames=$f: "
    KVZ_HIP_LIB=$PWD/$so timeout 300 python bench.py --frames $f --steps 3 --warmup 1 --no-cpu-baseline --no-ref-encoder 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
