#!/usr/bin/env python3
"""Developer tool (GPU box): bench.py's 1920x1080 like-for-like chain (two batches of 1 536 pictures in turn, every batch uploaded from pinned host memory inside the timed
region) over and over, each attempt's wall time and the library's messages -- what a failed or slow attempt looked like.  usage: tools/chain_stress.py [rounds=20] [pictures=1536] [up|resident] [swap|notfirst]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model, pinned_bytes
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
uploads = (sys.argv[3] if len(sys.argv) > 3 else "up") == "up"
w, h = 1920, 1080
lib = kvazaar_amd.load_library()
model = cost_model(lib, 22)
frames = bench.synth_frames(w, h, 4, bench.clip_seed(w, h))
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "entropy.json"))).get("bench-1080p")
dummy = HipBatch(lib, w, h, 8) if len(sys.argv) > 4 and sys.argv[4] == "notfirst" else None  # neither of the two is the process's first batch (first stream, first allocations)
pair = [HipBatch(lib, w, h, n) for _ in range(2)]
if len(sys.argv) > 4 and sys.argv[4] == "swap":
    pair = pair[::-1]  # the batch created second takes the first turn
for b in pair:
    for i in range(n):
        b.upload(i, frames[i % 4])
fb = w * h * 3 // 2
src_ptr, view = pinned_bytes(lib, n * fb)
for i in range(n):
    view[i * fb:(i + 1) * fb] = frames[i % 4]
for r in range(rounds):
    t = time.perf_counter()
    try:
        s, pics, per, ok = bench.chain_full(pair, model, 22, 3, gold, 4, src_ptr=src_ptr)[:4] if uploads else bench.chain_full(pair, model, 22, 3, gold, 4)
        print(r, f"{s / 6 * 1e3:.1f} ms per batch, attempt {time.perf_counter() - t:.2f} s, load {os.getloadavg()[0]:.1f}", ok, flush=True)
    except Exception as e:
        print(r, f"FAILED after {time.perf_counter() - t:.2f} s, load {os.getloadavg()[0]:.1f}", repr(e), flush=True)
        import numpy as np, ctypes
        lib.kvz_hip_batch_debug_flags.restype = ctypes.c_uint
        lib.kvz_hip_batch_debug_flags.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        for k, b in enumerate(pair):
            flags = np.zeros((b.n, 17, 30), np.uint32)
            ep = lib.kvz_hip_batch_debug_flags(b.handle, flags.ctypes.data)
            open_frames = [f for f in range(b.n) if (flags[f] != ep).any()]
            print(f"   batch {k}: last pass {ep}; pictures with CTUs not completed by it: {len(open_frames)} {open_frames[:12]}", flush=True)
            for f in open_frames[:2]:
                print(f"   picture {f}: per CTU row, the CTUs completed by pass {ep} (#) / by an earlier pass (.)", flush=True)
                for y in range(17):
                    print("     " + "".join("#" if flags[f, y, x] == ep else "." for x in range(30)), flush=True)
        if hasattr(lib, "kvz_hip_batch_debug_trace"):  # a -DKVZ_CTU_TRACE build: where every CTU of the failed pass stood when the first wait gave up
            lib.kvz_hip_batch_debug_trace.restype = None
            lib.kvz_hip_batch_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            names = {0: "untouched", 1: "ticket drawn / waiting for neighbours", 2: "neighbours there: searching", 3: "searched", 4: "stores drained", 5: "released", 6: "flag stored"}
            for k, b in enumerate(pair):
                tr = np.zeros((b.n, 17, 30), np.uint64)
                lib.kvz_hip_batch_debug_trace(b.handle, tr.ctypes.data)
                ep = int((tr >> np.uint64(8)).max())
                cur = (tr >> np.uint64(8)) == np.uint64(ep)
                st = (tr & np.uint64(255)).astype(int)
                print(f"   batch {k}: trace of pass {ep}: " + ", ".join(f"{names[v]}: {int((cur & (st == v)).sum())}" for v in range(1, 7)) + f"; CTUs still at an earlier pass: {int((~cur).sum())}", flush=True)
                for v in (2, 3, 4, 5):
                    for f, y, x in np.argwhere(cur & (st == v))[:40]:
                        print(f"      {names[v]}: picture {f} x {x} y {y}", flush=True)
        for b in pair:
            b.reset()
