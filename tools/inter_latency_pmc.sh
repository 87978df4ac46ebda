#!/bin/bash
# On the GPU box: the memory latencies a wavefront of the inter CTU pass sees (SQ_INST_LEVEL_* / SQ_INSTS_* = average cycles an access is in flight) and what its
# cycles wait on, for the default library and the variants named.  rocprofv3 --pmc passes only.  usage: tools/inter_latency_pmc.sh <tag> [variant...]
tag=$1; shift
repo=$PWD
cd /tmp && export TMPDIR=/tmp
P="python $repo/tools/inter_ctu_probe.py survey-416x240 1024"
for v in default "$@"; do
  if [ $v != default ]; then export KVZ_HIP_LIB=$repo/kvazaar_amd/lib/variants/libkvz_hip_$v.so; fi
  run() { timeout 300 rocprofv3 --pmc "${@:2}" --output-format csv -d $repo/gpurun_out/${tag}_lat_${v}_$1 -- $P > $repo/gpurun_out/${tag}_lat_${v}_$1.log 2>&1; }
  run a SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run b SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run c SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU
  run d SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_WAVES SQ_INST_CYCLES_VMEM_RD
done
cd $repo
python - <<PY
import csv, glob, collections, json, re
res = {}
for p in sorted(glob.glob("gpurun_out/${tag}_lat_*/**/*counter_collection.csv", recursive=True)):
    v = re.search(r"_lat_(.+)_[a-d]/", p).group(1)
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        if "inter_ctu" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: res.setdefault(v, {})[k] = acc[k] / n[k]
for v, c in res.items():
    print(v)
    for k in sorted(c): print(f"  {k:26s} {c[k]:.5g}")
json.dump(res, open("gpurun_out/${tag}_inter_latency_pmc.json", "w"), indent=1)
PY
