#!/usr/bin/env python3
"""Developer tool (GPU box): frames per second of the REAL encoder -- oracle/_ref/kvazaar_ref with its AVX2 strategies against
oracle/_ref/kvazaar_hip with the device searching gathered pictures (integration/kvazaar/search_lcu_hip.c, KVZ_HIP_BATCH_SEARCH=1) -- on the
benchmark's 1080p clip, all-intra, same options; also checks that the two bitstreams are identical.  usage: tools/encoder_fps.py [frames] [preset] [qp]"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvazaar_amd import synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
preset = sys.argv[2] if len(sys.argv) > 2 else "ultrafast"
qp = sys.argv[3] if len(sys.argv) > 3 else "22"
w, h = 1920, 1080
yuv = "/tmp/enc_fps.yuv"
distinct = [b"".join(p.tobytes() for p in planes) for planes in synth.frames(w, h, 8, 1, "large")]
with open(yuv, "wb") as f:
    for i in range(frames):
        f.write(distinct[i % 8])
threads = len(os.sched_getaffinity(0))
try:  # the CPUs the container is granted (cgroup v2), not the ones it can see: both encoders get that many threads
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    if quota != "max":
        threads = max(1, min(threads, int(int(quota) / int(period))))
except (OSError, ValueError):
    pass


def run(binary, extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = f"/tmp/enc_fps_{binary}.hevc"
    cmd = [os.path.join(ROOT, "oracle", "_ref", binary), "-i", yuv, "--input-res", f"{w}x{h}", "--preset", preset, "-p", "1", "-q", qp, "-o", out] + extra
    t = time.time()
    r = subprocess.run(cmd, env=e, capture_output=True, text=True)
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-1500:]
    return frames / dt, hashlib.md5(open(out, "rb").read()).hexdigest()


for owf in (15, 63):
    opts = ["--threads", str(threads), "--owf", str(owf)]
    fps_ref, md5_ref = run("kvazaar_ref", opts)
    trace = "/tmp/enc_fps_trace"
    fps_hip, md5_hip = run("kvazaar_hip", opts, {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": trace})
    etrace = "/tmp/enc_fps_etrace"
    if os.path.exists(etrace):
        os.remove(etrace)
    fps_ent, md5_ent = run("kvazaar_hip", opts, {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_ENTROPY": "1", "KVZ_HIP_ENTROPY_TRACE": etrace})
    coded = open(etrace).read().strip() if os.path.exists(etrace) else "0"
    print(f"preset {preset} QP {qp} {frames} frames --threads {threads} --owf {owf}: AVX2 encoder {fps_ref:.1f} fps | device search {fps_hip:.1f} fps "
          f"(pictures passes largest-batch: {open(trace).read().strip()}) | device search + device entropy coding {fps_ent:.1f} fps ({coded} pictures coded on the device) | "
          f"bitstreams identical: {md5_ref == md5_hip and md5_ref == md5_ent}", flush=True)
