#!/usr/bin/env python3
"""Fuzz of the entropy coder's device sources (host simulation, tests/hostsim) against the oracle's coder: random CU quadtrees (all depths, NxN CUs, all 35 modes so that
every scan order occurs), random levels from sparse to dense and from +-1 to +-32767 (escape codes, carries, emulation prevention), random SAO decisions, random QPs
(initial context states), WPP and --no-wpp, picture sizes that cut CTUs.  usage: tools/fuzz_entropy.py [rounds] [seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flatapi, ctu_common as cc, entropy_common as ec
from test_encoder_parity import oracle_model

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = flatapi.load_oracle()
sim = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libkvz_hostsim.so"))
f = sim.kvz_hostsim_entropy_code
f.restype = C.c_long
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]


def quadtree(w, h, nxn):
    depth = np.zeros((h // 8, w // 8), np.uint8)
    part = np.zeros((h // 8, w // 8), np.uint8)
    def split(x, y, d):
        size = 64 >> d
        if x >= w or y >= h:
            return
        must = x + size > w or y + size > h
        if d < 3 and (must or rng.random() < (0.75, 0.6, 0.45)[d]):
            for q in range(4):
                split(x + (q & 1) * size // 2, y + (q >> 1) * size // 2, d + 1)
            return
        depth[y // 8:(y + size) // 8, x // 8:(x + size) // 8] = d
        if d == 3 and nxn and rng.random() < 0.4:
            part[y // 8, x // 8] = 1
    for cy in range(0, h, 64):
        for cx in range(0, w, 64):
            split(cx, cy, 0)
    return depth, part


bad = 0
for r in range(rounds):
    w, h = int(rng.choice([64, 72, 128, 200, 264])), int(rng.choice([64, 88, 136, 192]))
    qp, nxn, sao_on, no_wpp = int(rng.integers(0, 52)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    model = oracle_model(oracle, qp)
    model.no_wpp = int(no_wpp)
    model.search_nxn = int(nxn)
    depth, part = quadtree(w, h, nxn)
    mode8 = rng.integers(0, 35, depth.shape).astype(np.uint8)
    mode4 = np.repeat(np.repeat(mode8, 2, 0), 2, 1).copy()
    m = np.repeat(np.repeat(part, 2, 0), 2, 1).astype(bool)
    mode4[m] = rng.integers(0, 35, int(m.sum()))
    mode8 = mode4[::2, ::2].copy()  # cu_mode holds the first PU's mode
    # every unit of a CU carries the CU's mode
    for y8 in range(depth.shape[0]):
        for x8 in range(depth.shape[1]):
            d = int(depth[y8, x8]); s8 = (64 >> d) // 8
            oy, ox = (y8 // s8) * s8, (x8 // s8) * s8
            if not part[y8, x8]:
                mode8[y8, x8] = mode8[oy, ox]
                mode4[2 * y8:2 * y8 + 2, 2 * x8:2 * x8 + 2] = mode8[oy, ox]
    wc, hc = (w + 63) // 64, (h + 63) // 64
    style = rng.integers(0, 4)
    coeff = np.zeros((wc * hc, 6144), np.int16)
    density = (0.01, 0.08, 0.4, 0.9)[style]
    mag = (2, 6, 300, 32767)[int(rng.integers(0, 4))]
    mask = rng.random(coeff.shape) < density
    coeff[mask] = rng.integers(-mag, mag + 1, int(mask.sum()))
    o = {"depth": np.ascontiguousarray(depth.reshape(-1)), "mode": np.ascontiguousarray(mode8.reshape(-1)), "coeff": np.ascontiguousarray(coeff.reshape(-1))}
    if nxn:
        o["part"], o["mode4"] = np.ascontiguousarray(part.reshape(-1)), np.ascontiguousarray(mode4.reshape(-1))
    sao = None
    recs = merge = None
    if sao_on:
        n = wc * hc
        lum, chr_ = np.zeros((n, 15), np.int32), np.zeros((n, 15), np.int32)
        for arr in (lum, chr_):
            arr[:, 0] = rng.integers(0, 3, n); arr[:, 1] = rng.integers(0, 4, n); arr[:, 2:4] = rng.integers(0, 32, (n, 2)); arr[:, 14] = 8
            arr[:, 4:14] = rng.integers(-7, 8, (n, 10))
            edge = arr[:, 0] == 2  # edge offsets: categories 1, 2 >= 0 and 3, 4 <= 0 (only |offset| is coded)
            for base in (4, 9):
                arr[edge, base + 1:base + 3] = np.abs(arr[edge, base + 1:base + 3]); arr[edge, base + 3:base + 5] = -np.abs(arr[edge, base + 3:base + 5])
        merge = rng.integers(0, 3, n).astype(np.uint8)
        for i in range(n):
            lx, ly = i % wc, i // wc
            if merge[i] == 1 and lx == 0: merge[i] = 0
            if merge[i] == 2 and ly == 0: merge[i] = 0
        sao = (np.ascontiguousarray(lum).view(np.uint8).reshape(-1), np.ascontiguousarray(chr_).view(np.uint8).reshape(-1), merge)
        recs = ec.pack_sao_records(sao[0], sao[1], n)
    want, want_sizes = ec.oracle_entropy(oracle, model, w, h, o, sao)
    buf, sizes, most = np.zeros(len(want) + 65536, np.uint8), np.zeros(hc, np.uint32), C.c_uint32(0)
    total = f(C.addressof(model), w, h, 1, o["depth"].ctypes.data, o["mode"].ctypes.data, o["part"].ctypes.data if nxn else None, o["mode4"].ctypes.data if nxn else None,
              o["coeff"].ctypes.data, recs.ctypes.data if recs is not None else None, merge.ctypes.data if merge is not None else None, 65536, buf.ctypes.data, sizes.ctypes.data, C.byref(most))
    ok = total == len(want) and buf[:total].tobytes() == want and [int(v) for v in sizes[:1 if no_wpp else hc]] == want_sizes
    bad += not ok
    print(f"round {r}: {w}x{h} qp {qp} nxn {int(nxn)} sao {int(sao_on)} no_wpp {int(no_wpp)} density {density} mag {mag}: {len(want)} bytes, most records {most.value}: {'ok' if ok else 'DIFFERENT'}", flush=True)
print("differences:", bad)
sys.exit(1 if bad else 0)
