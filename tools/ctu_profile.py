#!/usr/bin/env python3
"""Developer tool: per-stage cycle breakdown of the CTU kernel.  Builds a -DKVZ_CTU_PROFILE copy of the library into
gpurun_out/, runs one 1080p batch and prints the share of each stage (lane-0 shader clock, summed over workgroups)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["init", "refs", "pred35+replay", "satd", "select", "recon_pred", "fdct", "quant", "idct", "recon", "cost", "copy", "finish", "misc", "coeffbits", "rdoq"]

RDOQ_SECTIONS = ["setup / last position / final", "group: positions", "group: decisions per class", "group: chain walk", "group: costs", "group: ordered sums", "group: decision, levels out; last pass", "the calls as the caller sees them (routine + call overhead)"]


def main():
    out = os.path.join(ROOT, "kvazaar_amd", "lib", "variants", "libkvz_hip_prof.so")  # built ahead (cross-compiles without a GPU): python tools/ctu_profile.py --build
    if "--build" in sys.argv or not os.path.exists(out):
        from kvazaar_amd import build
        build.build_variant("prof", ["-DKVZ_CTU_PROFILE", "-DKVZ_RDOQ_WAVES_PER_EU=3"] + os.environ.get("KVZ_PROFILE_FLAGS", "").split())  # the timers cost registers: 3 wavefronts per SIMD there
        if "--build" in sys.argv:
            return
    import numpy as np
    import ctu_common as cc
    import bench
    lib = C.CDLL(out)
    qp = int(os.environ.get("KVZ_PROFILE_QP", "22"))
    model = cc.hip_cost_model(lib, qp, cc.coeff_weights(qp))
    if os.environ.get("KVZ_PROFILE_CABAC"):   # fast-residual-cost 0 (presets faster and up)
        model.coeff_cabac = 1
    if os.environ.get("KVZ_PROFILE_S32"):     # --pu-depth-intra 1-3 (preset fast)
        model.search_32x32 = 1
    if os.environ.get("KVZ_PROFILE_RDOQ"):    # --rdoq (preset medium without NxN)
        model.coeff_cabac = model.search_32x32 = model.rdoq = 1
    if os.environ.get("KVZ_PROFILE_NXN"):     # + the NxN partition: preset medium
        model.search_nxn = 1
    frames = bench.synth_frames(1920, 1080, 4, 1)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    b = cc.HipBatch(lib, 1920, 1080, n)
    for i in range(n):
        b.upload(i, frames[i % len(frames)])
    b.run(model)
    NW = 4 * len(NAMES) + 16
    buf = (C.c_ulonglong * NW)()
    n_cat = len(NAMES)
    lib.kvz_hip_batch_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    lib.kvz_hip_batch_profile(b.handle, buf, NW)
    b.run(model)
    lib.kvz_hip_batch_profile(b.handle, buf, NW)
    tot = sum(buf[:len(NAMES)])
    tot_all = tot + sum(buf[2 * len(NAMES) + 16:3 * len(NAMES) + 16])
    nctu = n * 510
    print(f"kernel_ms {b.kernel_ms():.2f}  cycles/CTU {tot_all / nctu:.0f} (outside the 4x4 PUs {tot / nctu:.0f})")
    for i, nm in enumerate(NAMES):
        print(f"{nm:11s} {buf[i] / nctu:10.0f} cyc/CTU  {100.0 * buf[i] / tot_all:5.1f}%  {buf[len(NAMES) + i] / nctu:7.1f} marks/CTU")
    rq = [buf[2 * len(NAMES) + i] for i in range(8)]
    if sum(rq):
        print("rdoq_block_wave, luma blocks (cycles of the wavefront's own clock per CTU):")
        for i, nm in enumerate(RDOQ_SECTIONS):
            print(f"  {nm:28s} {rq[i] / nctu:10.0f}  {100.0 * rq[i] / sum(rq):5.1f}%")
        base = 2 * len(NAMES) + 8
        for k in range(4):
            calls = buf[base + 4 + k]
            if calls:
                print(f"  luma {4 << k:2d}x{4 << k:<2d} blocks: {calls / nctu:7.1f} per CTU, {buf[base + k] / calls:9.0f} cycles each, {buf[base + k] / nctu:10.0f} per CTU")
    pu = 2 * len(NAMES) + 16
    if sum(buf[pu:pu + len(NAMES)]):
        print("inside the 4x4 PUs of the NxN attempt (eval_pu; included in the table above? no: listed separately, the table above is the rest):")
        for i, nm in enumerate(NAMES):
            if buf[pu + i]:
                print(f"  {nm:11s} {buf[pu + i] / nctu:10.0f} cyc/CTU  {100.0 * buf[pu + i] / (tot + sum(buf[pu:pu + len(NAMES)])):5.1f}%  {buf[pu + len(NAMES) + i] / nctu:7.1f} marks/CTU")


if __name__ == "__main__":
    main()

