"""Helpers of the entropy-coding tests: an H.265 byte-stream splitter (start codes, emulation prevention), the expected tail of kvazaar's slice header (entry points)
and the ctypes call of the oracle's real-mode coder (oracle/kvz_oracle_entropy.inc)."""
import ctypes as C

import numpy as np

import ctu_common as cc
from flatapi import ptr


def without_emulation_prevention(data):
    body, z = bytearray(), 0
    for b in data:
        if z >= 2 and b == 3:
            z = 0
            continue
        body.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(body)


def nal_units(stream, raw=False):
    """[(nal_unit_type, bytes after the two-byte NAL header)] of an Annex B byte stream; emulation prevention bytes removed unless raw"""
    out, i, n = [], 0, len(stream)
    starts = []
    while i + 3 <= n:
        if stream[i] == 0 and stream[i + 1] == 0 and stream[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    for k, s in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else n
        while e > s and stream[e - 1] == 0:  # trailing zero_byte of the next start code prefix
            e -= 1
        nal = stream[s:e]
        out.append(((nal[0] >> 1) & 0x3F, bytes(nal[2:]) if raw else without_emulation_prevention(nal[2:])))
    return out


def slice_payloads(stream):
    """the payload of every VCL NAL unit (types 0..21) as it stands in the stream (emulation prevention bytes included), in order"""
    return [body for t, body in nal_units(stream, raw=True) if t <= 21]


def ue_bits(v):
    v += 1
    n = v.bit_length()
    return "0" * (n - 1) + format(v, "b")


def entry_point_bits(sizes):
    """what kvz_encoder_state_write_bitstream_slice_header writes last (encoder_state-bitstream.c:935-954): num_entry_point_offsets, offset_len_minus1, the offsets of
    all substreams but the last, then the header's rbsp_trailing_bits -- the caller pads the result to a byte boundary with zeros"""
    n = len(sizes) - 1
    s = ue_bits(n)
    if n > 0:
        ln = max(sizes).bit_length()
        s += ue_bits(ln - 1)
        for b in sizes[:-1]:
            s += format(b - 1, "0%db" % ln)
    return s + "1"


def header_ends_with_entry_points(header, sizes, wpp):
    """header: the bytes in front of the substreams as they stand in the NAL unit"""
    bits = "".join(format(b, "08b") for b in without_emulation_prevention(header))
    tail = entry_point_bits(sizes) if wpp else "1"
    bits = bits.rstrip("0")  # the alignment zeros
    return bits.endswith(tail.rstrip("0")) if tail.rstrip("0") else True


def oracle_entropy(oracle, model, width, height, o, sao=None, not_last=0):
    """kvz_oracle_entropy_intra_picture on the outputs `o` of a CTU pass (rec / coeff / depth / mode [/ part / mode4]); sao = (luma, chroma, merge) arrays or None.
    Returns (bytes of all substreams, [substream sizes])"""
    f = oracle.lib.kvz_oracle_entropy_intra_tile
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    hc = (height + 63) // 64
    cap = width * height * 12 + 65536  # dense blocks of 16-bit levels cost 5 bytes per sample and more
    out = np.zeros(cap, np.uint8)
    sizes = np.zeros(hc, np.uint32)
    part = o.get("part")
    mode4 = o.get("mode4")
    n = f(C.addressof(model), width, height, o["depth"].ctypes.data, o["mode"].ctypes.data, part.ctypes.data if part is not None else None,
          mode4.ctypes.data if mode4 is not None else None, o["coeff"].ctypes.data, sao[0].ctypes.data if sao else None, sao[1].ctypes.data if sao else None,
          sao[2].ctypes.data if sao else None, int(not_last), out.ctypes.data, cap, sizes.ctypes.data)
    assert n <= cap
    k = 1 if model.no_wpp else hc
    return out[:n].tobytes(), [int(v) for v in sizes[:k]]


# (name, width, height, frames, seed, kind, qp, preset, extra CLI options): all-intra (-p 1) encodes whose slice data the coder has to reproduce
CASES = [
    ("ultrafast-qp22", 416, 240, 3, 1234, "small", 22, "ultrafast", []),                 # BASELINE config 1's clip and options
    ("ultrafast-qp32", 416, 240, 2, 1234, "small", 32, "ultrafast", []),
    ("ultrafast-no-wpp", 416, 240, 2, 1234, "small", 22, "ultrafast", ["--no-wpp"]),     # one substream per picture
    ("partial-ctus-qp27", 200, 136, 2, 3, "small", 27, "ultrafast", []),                 # picture borders inside CTUs: implicit splits
    ("one-ctu-wide", 64, 200, 2, 9, "small", 22, "ultrafast", []),                       # rows never receive the upper row's contexts
    ("noise-qp12", 192, 136, 4, 0, "adversarial", 12, "ultrafast", []),                  # flat / noise / ramp / blocks: escape codes, empty CTUs
    ("noise-qp37", 192, 136, 4, 0, "adversarial", 37, "ultrafast", []),
    ("veryfast-sao", 416, 240, 2, 1234, "small", 22, "veryfast", []),                    # SAO syntax in front of every CTU
    ("veryfast-sao-noise", 192, 136, 4, 0, "adversarial", 22, "veryfast", []),           # band offsets, merges
    ("medium", 416, 240, 2, 1234, "small", 22, "medium", []),                            # BASELINE config 3's preset: RDOQ levels, NxN CUs, SAO
    ("medium-nxn-everywhere", 192, 136, 4, 0, "adversarial", 12, "medium", []),
    ("ultrafast-832x480", 832, 480, 1, 5, "large", 22, "ultrafast", []),
]

# the pictures bench.py keeps resident (the first 8 frames of SURVEY.md App. C's 1080p clip): fixture entries only -- the device's coder is checked against them on the GPU
# and by bench.py's entropy leg; the oracle's on the build machine when the fixture is made
BENCH_CASES = [("bench-1080p", 1920, 1080, 8, 1, "large", 22, "ultrafast", []),
               ("bench-2160p", 3840, 2160, 4, 2, "large", 22, "ultrafast", [])]  # ... and the four pictures of its 3840x2160 legs (App. C's 2160p clip)


def case_model(oracle, case):
    """the cost model the CTU pass of the case's preset runs with (what tests/test_encoder_parity.py pins against the reference encoder's reconstruction)"""
    from test_encoder_parity import _medium_model, oracle_model
    name, w, h, n, seed, kind, qp, preset, extra = case
    model = oracle_model(oracle, qp)
    if preset == "medium":
        model = _medium_model(model)
    if "--no-wpp" in extra:
        model.no_wpp = 1
    return model


def oracle_slice_data(oracle, case):
    """[(slice data, substream sizes)] per picture: the oracle's CTU pass (+ its SAO decision for presets with SAO), then its real-mode coder"""
    from test_sao_decision import oracle_sao_chain
    name, w, h, n, seed, kind, qp, preset, extra = case
    model = case_model(oracle, case)
    out = []
    for f in cc.yuv_frames(w, h, n, seed, kind):
        o = cc.run_oracle_nxn(oracle, model, w, h, f) if preset == "medium" else cc.run_oracle(oracle, model, w, h, f)
        sao = None
        if preset != "ultrafast":
            _, _, luma, chroma, merge, _ = oracle_sao_chain(oracle, model, w, h, f, pre=o)
            sao = (np.frombuffer(luma, np.uint8), np.frombuffer(chroma, np.uint8), merge)
        out.append(oracle_entropy(oracle, model, w, h, o, sao))
    return out


def reference_slice_payloads(ref_binary, case, workdir):
    """the slice NAL payloads kvazaar_ref writes for the case"""
    import os
    import subprocess
    from kvazaar_amd import synth
    name, w, h, n, seed, kind, qp, preset, extra = case
    yuv, out = os.path.join(workdir, "in.yuv"), os.path.join(workdir, "out.hevc")
    if kind == "adversarial":
        open(yuv, "wb").write(b"".join(f.tobytes() for f in cc.yuv_frames(w, h, n, seed, kind)))
    else:
        synth.write_yuv(yuv, w, h, n, seed, kind)
    r = subprocess.run([ref_binary, "-i", yuv, "--input-res", f"{w}x{h}", "--preset", preset, "-p", "1", "-q", str(qp), "-n", str(n), "--threads", "4", "-o", out] + extra,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    return slice_payloads(open(out, "rb").read())


def pack_sao_records(luma, chroma, n_lcus):
    """kvz_sao.hpp SaoRec per (LCU, plane) from kvz_hip_sao_params arrays (the layout the device's SAO decision writes and kvz_hip_batch_sao_params unpacks):
    type | eo_class << 8 | band_position << 16 | five offsets as signed bytes from bit 24"""
    lp = np.frombuffer(luma, np.int32).reshape(n_lcus, 15)   # type, eo_class, band_position[2], offsets[10], bitdepth
    cp = np.frombuffer(chroma, np.int32).reshape(n_lcus, 15)
    recs = np.zeros((n_lcus, 3), np.uint64)
    for i in range(n_lcus):
        for plane, (p, slot) in enumerate(((lp[i], 0), (cp[i], 0), (cp[i], 1))):
            r = int(p[0]) | int(p[1]) << 8 | int(p[2 + slot]) << 16
            for k in range(5):
                r |= (int(p[4 + 5 * slot + k]) & 0xFF) << (24 + 8 * k)
            recs[i, plane] = r
    return recs


def hostsim_slice_data(oracle, hostsim_cdll, case, cap=12288, retry=True):
    """the device's entropy coder (kvazaar_amd/csrc/kvz_entropy.hpp) compiled for the host, on the oracle's CTU-pass outputs: [(slice data, substream sizes)] per picture"""
    from test_sao_decision import oracle_sao_chain
    name, w, h, n, seed, kind, qp, preset, extra = case
    model = case_model(oracle, case)
    f = hostsim_cdll.kvz_hostsim_entropy_code
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    wc, hc = (w + 63) // 64, (h + 63) // 64
    out = []
    for fr in cc.yuv_frames(w, h, n, seed, kind):
        o = cc.run_oracle_nxn(oracle, model, w, h, fr) if preset == "medium" else cc.run_oracle(oracle, model, w, h, fr)
        recs = merge = None
        if preset != "ultrafast":
            _, _, luma, chroma, merge, _ = oracle_sao_chain(oracle, model, w, h, fr, pre=o)
            recs = pack_sao_records(luma, chroma, wc * hc)
        part, mode4 = o.get("part"), o.get("mode4")
        buf, sizes, most = np.zeros(w * h * 4 + 4096, np.uint8), np.zeros(hc, np.uint32), C.c_uint32(0)
        def run(cap):
            return f(C.addressof(model), w, h, 1, o["depth"].ctypes.data, o["mode"].ctypes.data, part.ctypes.data if part is not None else None,
                     mode4.ctypes.data if mode4 is not None else None, o["coeff"].ctypes.data, recs.ctypes.data if recs is not None else None,
                     merge.ctypes.data if merge is not None else None, cap, buf.ctypes.data, sizes.ctypes.data, C.byref(most))
        total = run(cap)
        if total == -1 and retry:  # a CTU's bin list did not fit: again with the capacity it needs (what kvz_hip_batch_entropy_code does)
            total = run(most.value)
        assert total >= 0, (total, most.value)
        out.append((buf[:total].tobytes(), [int(v) for v in sizes[:1 if model.no_wpp else hc]]))
    return out
