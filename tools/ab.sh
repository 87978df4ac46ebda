#!/bin/bash
# On the GPU box: alternate bench runs of library variants (same box, interleaved) -- usage: tools/ab.sh <reps> <frames> libA libB ...
cd "$(dirname "$0")/.."
reps=$1; frames=$2; shift 2
for r in $(seq $reps); do
  for so in "$@"; do
    echo -n "$(basename $so) rep$r: "
    KVZ_HIP_LIB=$PWD/$so timeout 300 python bench.py --frames $frames --steps 3 --warmup 1 --no-cpu-baseline --no-ref-encoder 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))"
  done
done
