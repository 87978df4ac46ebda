#!/usr/bin/env python3
"""Development probe: kvz_hip_batch_entropy_code on a resident batch of 1920x1080 pictures (the headline bench's workload), timed.
usage: tools/entropy_probe.py [pictures] [qp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qp = int(sys.argv[2]) if len(sys.argv) > 2 else 22
w, h = 1920, 1080
lib = kvazaar_amd.load_library()
frames = bench.synth_frames(w, h, 8, bench.clip_seed(w, h))
b = HipBatch(lib, w, h, n)
for i in range(n):
    b.upload(i, frames[i % 8])
m = cost_model(lib, qp)
b.run(m)
t = time.perf_counter(); b.run(m); pass_s = time.perf_counter() - t
data, sizes = b.entropy_code(m)  # warm-up (allocations)
best = 1e9
for _ in range(3):
    t = time.perf_counter(); data, sizes = b.entropy_code(m); best = min(best, time.perf_counter() - t)
ctus = b.ctus_per_frame * n
print("pictures %d QP %d: CTU pass %.1f ms (%.0f CTUs/s); entropy coding %.1f ms = %.0f pictures/s = %.2f M CTUs/s; %d bytes of slice data (%.1f KB per picture) instead of %.1f MB of levels per picture"
      % (n, qp, pass_s * 1e3, ctus / pass_s, best * 1e3, n / best, ctus / best / 1e6, len(data), len(data) / n / 1e3, b.ctus_per_frame * 12288 / 1e6))
same = all(np.array_equal(sizes[i], sizes[i % 8]) for i in range(n))
print("copies consistent:", same)
b.close()
