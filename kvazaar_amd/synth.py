"""Synthetic YUV generator of SURVEY.md App. C (numpy default_rng / PCG64), yuv420p 8-bit planar."""
import hashlib

import numpy as np

# md5 of the files SURVEY.md App. C records: 416x240 x 8 frames (seed 1234, "small"), 1920x1080 x 8 (seed 1, "large"), 3840x2160 x 4 (seed 2, "large")
MD5 = {"416x240": "c87920652c571cde553cb44f0d039522", "1920x1080": "4aa32ffcf953b3bcb14fd67659b963b1", "3840x2160": "7379f169ab620d1bbbef3255acb1ccb1"}


def frames(w, h, n, seed, kind):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        if kind == "small":
            Y = 128 + 60 * np.sin((xx + 3 * i) / 17) + 50 * np.cos((yy - 2 * i) / 11) + rng.normal(0, 6, (h, w))
            U = 128 + 30 * np.sin((xx[::2, ::2] + i) / 23)
            V = 128 + 30 * np.cos((yy[::2, ::2] + i) / 19)
        else:
            Y = 128 + 60 * np.sin((xx + 3 * i) / 37) + 50 * np.cos((yy - 2 * i) / 23) + 20 * np.sin(xx * yy / 9000) + rng.normal(0, 6, (h, w))
            U = 128 + 30 * np.sin((xx[::2, ::2] + i) / 43)
            V = 128 + 30 * np.cos((yy[::2, ::2] + i) / 39)
        yield tuple(np.clip(p, 0, 255).astype(np.uint8) for p in (Y, U, V))


def write_yuv(path, w, h, n, seed, kind):
    m = hashlib.md5()
    with open(path, "wb") as f:
        for planes in frames(w, h, n, seed, kind):
            for p in planes:
                b = p.tobytes()
                f.write(b)
                m.update(b)
    return m.hexdigest()
