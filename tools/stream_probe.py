#!/usr/bin/env python3
"""Developer tool: a few launches of one streaming kernel (for rocprofv3 --pmc passes).  usage: stream_probe.py sao|deblock [edge_class]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kvazaar_amd
from kvazaar_amd.dev import Dev
import sao_common as sc

def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "sao"
    dev = Dev(kvazaar_amd.load_library())
    w, h, nfr = 1920, 1080, 128
    rng = np.random.default_rng(1)
    frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    if what == "sao":
        typ, cls = (2, int(sys.argv[2])) if len(sys.argv) > 2 else (0, 0)
        nctu = 30 * 17
        lum, chr_ = sc.random_params(rng, nctu, False), sc.random_params(rng, nctu, True)
        for arr in (lum, chr_):
            for q in arr:
                q.type, q.eo_class = typ, cls
        dl = dev.put(np.tile(np.frombuffer(bytes(lum), dtype=np.uint8), nfr)); dch = dev.put(np.tile(np.frombuffer(bytes(chr_), dtype=np.uint8), nfr))
        din, dout = dev.put(np.tile(frame, (nfr, 1))), dev.empty(nfr * frame.nbytes)
        dev.lib.kvz_hip_dev_sao_frames.restype = None
        dev.lib.kvz_hip_dev_sao_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for _ in range(5):
            dev.lib.kvz_hip_dev_sao_frames(din, dout, w, h, nfr, dl, dch)
        dev.lib.kvz_hip_dev_sync()

if __name__ == "__main__":
    main()
