#!/bin/bash
# On the GPU box: `bench.py --only medium` (3840x2160 `--preset medium`, 48 pictures) through the default library and the named variants.  usage: tools/medium_variants.sh <tag> <name>...
tag=$1; shift
python bench.py --only medium --medium-frames ${MEDIUM_FRAMES:-48} --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default: %.0f CTUs/s kernel %.1f ms verified %s' % (d['value'], d['kernel_ms'], d['verified']))" | tee gpurun_out/${tag}_medium_variants.log
for v in "$@"; do
  KVZ_HIP_LIB=$PWD/kvazaar_amd/lib/variants/libkvz_hip_$v.so python bench.py --only medium --medium-frames ${MEDIUM_FRAMES:-48} --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.0f CTUs/s kernel %.1f ms verified %s' % (d['value'], d['kernel_ms'], d['verified']))" | tee -a gpurun_out/${tag}_medium_variants.log
done
