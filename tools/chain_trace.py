#!/usr/bin/env python3
"""On the GPU box, under `rocprofv3 --kernel-trace`: a few turns of bench.py's chain_full loop (the other batch's pass started by the coder), for the kernels' timeline.
usage: tools/chain_trace.py [pictures_per_batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model
import bench

half = int(sys.argv[1]) if len(sys.argv) > 1 else 768
w, h = 1920, 1080
lib = kvazaar_amd.load_library()
model = cost_model(lib, 22)
frames = bench.synth_frames(w, h, 4, bench.clip_seed(w, h))
pair = []
for _ in range(2):
    b = HipBatch(lib, w, h, half)
    for i in range(half):
        b.upload(i, frames[i % 4])
    pair.append(b)
for b in pair:
    b.launch(model); b.deblock(22, wait=False); b.entropy_code(model)
cur = pair[0]
cur.launch(model); cur.deblock(22, wait=False)
for i in range(4):
    nxt = pair[(i + 1) & 1]
    cur.entropy_code(model, then=(nxt, model) if i < 3 else None)
    if i < 3: nxt.deblock(22, wait=False)
    cur = nxt
