"""Per-coefficient scaling lists for the parity cases (test infrastructure): a numpy restatement of what kvazaar's encoder control holds when
`--scaling-list` is not `off` -- kvz_scalinglist_process / kvz_scalinglist_set (scalinglist.c:318-342 forward, :289-312 inverse, :375-425), which is what
kvz_quant / kvz_dequant index with [log2 - 2][list][qp % 6] (quant-generic.c:59-60, 312-314).  Pinned against the compiled reference's own tables in
tests/test_oracle_vs_ref.py::test_scaling_lists_match_reference (default lists and the custom set below), so the GPU tests can build the lists without the
reference being present.

Two list sets:
  "default"  --scaling-list default (encoder.c:270-273): H.265 tables 7-5 / 7-6, DC 16
  "custom"   a seeded non-flat set (values 13..255 -- below 13 the reference's int16 coeff_t cannot hold (quant_scale << 4) / value -- and its own DC terms)"""
import numpy as np

QUANT_SCALES = (26214, 23302, 20560, 18396, 16384, 14564)  # scalinglist.c:78
INV_QUANT_SCALES = (40, 45, 51, 57, 64, 72)                # scalinglist.c:79
SIZE_X = (4, 8, 16, 32)
LIST_NUM = (6, 6, 6, 2)                                    # scalinglist.c:42

# H.265 table 7-6 (scalinglist.c:54-76), raster order; table 7-5 (4x4) is flat 16
DEFAULT_INTRA_8X8 = np.array([16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22, 25, 29, 16, 16, 18, 21, 24, 27, 31, 36,
                              17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27, 35, 44, 54, 65, 21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115], np.int32)
DEFAULT_INTER_8X8 = np.array([16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24, 25, 28, 16, 17, 18, 20, 24, 25, 28, 33,
                              17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25, 28, 33, 41, 54, 20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91], np.int32)


def list_type(cu_is_intra, type_):
    """quant-generic.c:59: (block_type == CU_INTRA ? 0 : 3) + "\\0\\3\\1\\2"[type]; the 32x32 size aliases list 3 to list 1 (scalinglist.c:104-108)"""
    return (0 if cu_is_intra else 3) + (0, 3, 1, 2)[type_]


class ListSet:
    """coeff[size][list]: the 16 / 64 list entries as parsed; dc[size][list]: 0 = not given (kvz_scalinglist_set then takes 16)"""

    def __init__(self, name):
        self.name = name
        self.coeff = np.zeros((4, 6, 64), np.int32)
        self.dc = np.zeros((4, 6), np.int32)
        if name == "default":  # kvz_scalinglist_get_default (scalinglist.c:261-279)
            for size in range(4):
                for lst in range(LIST_NUM[size]):
                    if size == 0:
                        self.coeff[size, lst, :16] = 16
                    elif size in (1, 2):
                        self.coeff[size, lst] = DEFAULT_INTER_8X8 if lst > 2 else DEFAULT_INTRA_8X8
                    else:
                        self.coeff[size, lst] = DEFAULT_INTER_8X8 if lst > 0 else DEFAULT_INTRA_8X8
        elif name == "custom":
            rng = np.random.default_rng(20260930)
            for size in range(4):
                for lst in range(LIST_NUM[size]):
                    n = 16 if size == 0 else 64
                    base = 13 + (np.arange(n) // (4 if size == 0 else 8) + np.arange(n) % (4 if size == 0 else 8)) * (6 + lst)  # rising towards high frequencies
                    self.coeff[size, lst, :n] = np.clip(base + rng.integers(0, 40, n), 13, 255)
                    self.dc[size, lst] = int(rng.integers(13, 64)) if size >= 2 else self.coeff[size, lst, 0]  # scalinglist.c:244-252: sizes below 16x16 take entry 0
        else:
            raise ValueError(name)

    def tables(self, log2w, lst, qp_rem):
        """(quant_coeff, de_quant_coeff) of width 2^log2w, int16 as the reference's coeff_t (kvz_scalinglist_set, scalinglist.c:375-391)"""
        size = log2w - 2
        if size == 3 and lst == 3:
            lst = 1
        w = SIZE_X[size]
        num = min(8, w)
        ratio = w // num
        dc = int(self.dc[size, lst]) or 16
        c = self.coeff[size, lst]
        j, i = np.mgrid[0:w, 0:w]
        pos = num * (j // ratio) + i // ratio
        div = np.where(pos > 63, 1, c[np.minimum(pos, 63)])
        quant = (QUANT_SCALES[qp_rem] << 4) // div           # kvz_scalinglist_process_enc
        deq = INV_QUANT_SCALES[qp_rem] * c[pos]              # scalinglist_process_dec
        if ratio > 1:
            quant[0, 0] = (QUANT_SCALES[qp_rem] << 4) // dc
            deq[0, 0] = INV_QUANT_SCALES[qp_rem] * dc
        return np.ascontiguousarray(quant.astype(np.int16).ravel()), np.ascontiguousarray(deq.astype(np.int16).ravel())

    def ref_args(self):
        """what oracle/ref_shim.c kvz_ref_set_scaling_list takes: mode, coeff[4][6][64] int16, dc[4][6] int32"""
        return (1 if self.name == "default" else 2), np.ascontiguousarray(self.coeff.astype(np.int16).ravel()), np.ascontiguousarray(self.dc.ravel())


_sets = {}


def get(name):
    if name not in _sets:
        _sets[name] = ListSet(name)
    return _sets[name]
