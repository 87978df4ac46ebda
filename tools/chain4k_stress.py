#!/usr/bin/env python3
"""Developer tool (GPU box): the 3840x2160 like-for-like chain of bench.py (two batches of 192 pictures in turn) over and over, with the library's messages on stderr.
usage: tools/chain4k_stress.py [rounds=10]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w, h, n = 3840, 2160, 192
lib = kvazaar_amd.load_library()
model = cost_model(lib, 22)
frames = bench.synth_frames(w, h, 4, 2)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "entropy.json"))).get("bench-2160p")
pair = [HipBatch(lib, w, h, n) for _ in range(2)]
for b in pair:
    for i in range(n):
        b.upload(i, frames[i % 4])
for r in range(rounds):
    try:
        s, pics, per, ok = bench.chain_full(pair, model, 22, 3, gold, 4)
        print(r, f"{s / 6 * 1e3:.1f} ms per batch", ok, flush=True)
    except Exception as e:
        print(r, "FAILED", repr(e), flush=True)
