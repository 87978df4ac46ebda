#!/usr/bin/env python3
"""Developer tool (GPU): does a device-to-host copy on one batch's stream overlap the CTU kernel of another batch?
Times the kernel alone, the copy alone and both together.  Run under different HSA_ENABLE_SDMA settings to see which copy path
(SDMA engines vs. blit kernels that need free wave slots) the runtime takes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import kvazaar_amd  # noqa: E402
from kvazaar_amd import synth  # noqa: E402
from kvazaar_amd.batch import HipBatch, PinnedResults, cost_model  # noqa: E402

print({k: v for k, v in os.environ.items() if k.startswith(("HSA", "HIP", "GPU_", "ROC"))})
lib = kvazaar_amd.load_library()
w, h, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 384
frames = [np.concatenate([p.reshape(-1) for p in pl]) for pl in synth.frames(w, h, 4, 1, "large")]
model = cost_model(lib, 22)
A, B = HipBatch(lib, w, h, n), HipBatch(lib, w, h, n)
for b in (A, B):
    for i in range(n):
        b.upload(i, frames[i % 4])
pa, pb = PinnedResults(A), PinnedResults(B)
A.run(model); B.run(model)
pb.download_async(); B.sync()  # first touch


def timed(f):
    t = time.perf_counter(); f(); return (time.perf_counter() - t) * 1e3


tk = timed(lambda: A.run(model))
tc = timed(lambda: (pb.download_async(), B.sync()))
both = timed(lambda: (A.launch(model), pb.download_async(), A.sync(), B.sync()))
rev = timed(lambda: (pb.download_async(), A.launch(model), A.sync(), B.sync()))
print(f"kernel {tk:.1f} ms | copy {tc:.1f} ms = {pb.bytes / tc / 1e6:.1f} GB/s | kernel then copy issued: {both:.1f} ms | copy then kernel issued: {rev:.1f} ms "
      f"(overlap => ~{max(tk, tc):.0f}, serial => ~{tk + tc:.0f})")
