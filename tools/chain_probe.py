#!/usr/bin/env python3
"""On the GPU box: timing forms of the all-intra chain (pass -> deblocking -> entropy coder -> slice data on the host) for bench.py's chain_full leg.
usage: tools/chain_probe.py [width height pictures_per_batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kvazaar_amd
from kvazaar_amd.batch import HipBatch, cost_model
import bench

w, h, half = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1920, 1080, 384)))
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

def stamp(name, fn, reps=3):
    for b in pair: b.sync()
    t = time.perf_counter(); fn(reps); dt = time.perf_counter() - t
    n = reps * 2 * half
    print(f"{name:34s} {dt / (2 * reps) * 1e3:8.1f} ms per batch   {n * pair[0].ctus_per_frame / dt / 1e3:8.1f} k CTUs/s   {n / dt:7.1f} pictures/s", flush=True)

def serial(reps):
    for _ in range(reps):
        for b in pair:
            b.launch(model); b.deblock(22, wait=False); b.entropy_code(model)
def interleaved(reps):
    pending = None
    for _ in range(reps):
        for b in pair:
            b.launch(model); b.deblock(22, wait=False)
            if pending is not None: pending.entropy_code(model)
            pending = b
    pending.entropy_code(model)
def pass_only(reps):
    for _ in range(reps):
        for b in pair:
            b.launch(model); b.deblock(22, wait=False); b.sync()
def entropy_only(reps):
    for _ in range(reps):
        for b in pair:
            b.entropy_code(model)
def parts(reps):
    for _ in range(reps):
        for b in pair:
            t0 = time.perf_counter(); b.launch(model); b.sync(); t1 = time.perf_counter(); b.deblock(22); t2 = time.perf_counter(); b.entropy_code(model); t3 = time.perf_counter()
    print(f"   last batch: pass {1e3 * (t1 - t0):.1f} ms, deblocking {1e3 * (t2 - t1):.1f} ms, entropy coder + download {1e3 * (t3 - t2):.1f} ms")
stamp("pass + deblocking", pass_only)
stamp("entropy coder + download", entropy_only)
stamp("serial chain", serial)
stamp("interleaved (bench.py chain_full)", interleaved)
stamp("serial, parts", parts)
import threading
def threaded(reps):
    # the coder's blocking call of one batch on a worker thread (ctypes releases the GIL), the other batch's pass queued from this thread a moment later
    worker = None
    for _ in range(reps):
        for b in pair:
            if worker is not None:
                time.sleep(0.003)
            b.launch(model); b.deblock(22, wait=False)
            if worker is not None:
                worker.join()
            worker = threading.Thread(target=b.entropy_code, args=(model,)); worker.start()
    worker.join()
stamp("threaded overlap", threaded)
def then(reps):
    # the other batch's pass started by the coder itself when its third stage is queued (kvz_hip_batch_entropy_code_then): bench.py's chain_full
    turns = 2 * reps
    cur = pair[0]
    cur.launch(model); cur.deblock(22, wait=False)
    for i in range(turns):
        nxt = pair[(i + 1) & 1]
        more = i + 1 < turns
        cur.entropy_code(model, then=(nxt, model) if more else None)
        if more: nxt.deblock(22, wait=False)
        cur = nxt
stamp("pass started by the coder", then)
for share in ((7, 8), (15, 16)):
    for b in pair: b.set_device_share(*share)
    stamp(f"... the pass on {share[0]}/{share[1]} of the slots", then)
for b in pair: b.set_device_share(1, 1)
