#!/bin/bash
# On the GPU box: bench.py's 4K inter leg at several batch sizes (sequences per launch).  usage: tools/inter_leg_sizes.sh <tag> <n>...
tag=$1; shift
for n in "$@"; do
  s=$(date +%s)
  python bench.py --only inter --inter-sequences $n > gpurun_out/${tag}_inter_leg_$n.json 2> gpurun_out/${tag}_inter_leg_$n.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_inter_leg_$n.json"))
print("$n sequences:", round(d["value"]), "CTUs/s,", round(d["ms"], 1), "ms per launch,", d["units_per_launch"], "CTUs, verified", d["verified"], ", wall", $(date +%s) - $s, "s")
PY
done
