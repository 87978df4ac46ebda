#!/usr/bin/env python3
"""Developer tool (GPU box): bench.py's chain_full with and without the uploads inside the timed region, alternately.  usage: tools/chain_h2d_ab.py [frames=1536] [rounds=2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model, pinned_bytes
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2  # batches in turn
w, h = 1920, 1080
lib = kvazaar_amd.load_library()
model = cost_model(lib, 22)
frames = bench.synth_frames(w, h, 8, 1)
fb = w * h * 3 // 2
pair = [HipBatch(lib, w, h, n) for _ in range(nb)]
for b in pair:
    for i in range(n):
        b.upload(i, frames[i % 8])
ptr, view = pinned_bytes(lib, n * fb)
for i in range(n):
    view[i * fb:(i + 1) * fb] = frames[i % 8]
for r in range(rounds):
    s0, pics, _, _ = bench.chain_full(pair, model, 22, 3, None, 8)
    s1, pics, _, _, ups = bench.chain_full(pair, model, 22, 3, None, 8, src_ptr=ptr)
    print(f"{nb} batches in turn: resident {s0 / (3 * nb) * 1e3:.1f} ms per batch | with uploads {s1 / (3 * nb) * 1e3:.1f} ms per batch ({ups} uploads)", flush=True)
