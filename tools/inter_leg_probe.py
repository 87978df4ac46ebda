#!/usr/bin/env python3
"""Development probe: bench.py's BASELINE config 4 leg on its own (I picture on the device, then the B pictures of many sequences chained through the loop filters).
usage: tools/inter_leg_probe.py [sequences]"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model

lib = kvazaar_amd.load_library()
args = types.SimpleNamespace(qp=22, no_ref_encoder=True, no_cpu_baseline=True)
d4 = bench.synth_frames(3840, 2160, 4, bench.clip_seed(3840, 2160))
print(json.dumps(bench.inter_leg(args, lib, lambda qp, tiles=None: cost_model(lib, qp), HipBatch, d4, int(sys.argv[1]) if len(sys.argv) > 1 else 192), indent=1))
