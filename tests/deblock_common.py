"""Shared helpers of the deblocking tests: random CU quadtrees, pictures that exercise every filter branch, ctypes calls."""
import ctypes as C

import numpy as np

from flatapi import ptr


def random_depth_map(width, height, rng):
    """CU depth per 8x8 unit from a random quadtree per CTU (depths 0..3), as the CTU pass returns it"""
    w8, h8 = width // 8, height // 8
    d = np.zeros((h8, w8), np.uint8)

    def split(x, y, size, depth):
        if depth < 3 and rng.random() < (0.75, 0.6, 0.5)[depth]:
            for dy in (0, size // 2):
                for dx in (0, size // 2):
                    split(x + dx, y + dy, size // 2, depth + 1)
        else:
            d[y:y + size, x:x + size] = depth

    for cy in range(0, h8, 8):
        for cx in range(0, w8, 8):
            split(cx, cy, 8, 0)
    return d[:h8, :w8].copy()


def test_picture(width, height, rng, kind):
    """planar 4:2:0 picture: 'smooth' (strong filter), 'steps' (weak filter around block edges), 'noise' (mostly no filtering)"""
    n = width * height
    yy, xx = np.mgrid[0:height, 0:width]
    if kind == "smooth":
        y = (xx // 8 * 3 + yy // 8 * 2 + 40) % 256
    elif kind == "steps":
        y = ((xx // 8 + yy // 8) % 2) * rng.integers(2, 12) + 100 + rng.integers(-1, 2, (height, width))
    else:
        y = rng.integers(0, 256, (height, width))
    c = rng.integers(0, 256, (2, height // 2, width // 2)) if kind == "noise" else \
        np.stack([(xx[::2, ::2] // 8 * 5 + 60) % 256, (yy[::2, ::2] // 8 * 7 + 90) % 256])
    return np.concatenate([np.clip(y, 0, 255).astype(np.uint8).reshape(-1), c.astype(np.uint8).reshape(-1)]), n


def run_cpu(lib_func, width, height, qp, beta_off, tc_off, frame, depth):
    out = frame.copy()
    n, c = width * height, width * height // 4
    lib_func.restype = None
    lib_func.argtypes = [C.c_int] * 5 + [C.c_void_p] * 4
    lib_func(width, height, qp, beta_off, tc_off, out.ctypes.data, out.ctypes.data + n, out.ctypes.data + n + c, np.ascontiguousarray(depth).ctypes.data)
    return out
