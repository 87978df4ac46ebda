#!/usr/bin/env python3
"""Developer tool (GPU box): what the asynchronous upload of a batch's pictures costs -- the copy alone, a pass alone, a pass with another batch's upload beside it.
usage: tools/h2d_probe.py [frames=1536]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model, pinned_bytes
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
w, h = 1920, 1080
lib = kvazaar_amd.load_library()
model = cost_model(lib, 22)
frames = bench.synth_frames(w, h, 8, 1)
fb = w * h * 3 // 2
a, b = HipBatch(lib, w, h, n), HipBatch(lib, w, h, n)
for i in range(n):
    a.upload(i, frames[i % 8]); b.upload(i, frames[i % 8])
ptr, view = pinned_bytes(lib, n * fb)
for i in range(n):
    view[i * fb:(i + 1) * fb] = frames[i % 8]
a.run(model); b.run(model)
for rep in range(2):
    t = time.perf_counter(); b.upload_all_async(ptr); b.run(model); s_up_pass = time.perf_counter() - t   # copy, then the pass that waits for it
    t = time.perf_counter(); a.run(model); s_pass = time.perf_counter() - t
    t = time.perf_counter(); b.upload_all_async(ptr); a.run(model); s_both = time.perf_counter() - t; k_both = a.kernel_ms()
    t = time.perf_counter(); b.run(model); s_after = time.perf_counter() - t  # waits for the rest of the copy, if any
    print(f"pass alone {s_pass * 1e3:.1f} ms | copy then pass {s_up_pass * 1e3:.1f} ms (copy ~{(s_up_pass - s_pass) * 1e3:.1f} ms = {n * fb / max(1e-9, s_up_pass - s_pass) / 1e9:.1f} GB/s) | "
          f"pass beside a copy {s_both * 1e3:.1f} ms (kernel {k_both:.1f}) | the copied batch's pass after it {s_after * 1e3:.1f} ms", flush=True)
