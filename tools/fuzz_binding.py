#!/usr/bin/env python3
"""Fuzz of the kvazaar-side binding (integration/kvazaar/search_lcu_hip.c) without a GPU: oracle/_ref/kvazaar_hipsim -- the integrated encoder with the batch calls served
by the oracle -- against oracle/_ref/kvazaar_ref on random command lines: every preset, all-intra (-p 1) or a low-delay GOP, --qp, --owf, --no-wpp, tiles, --sao, --no-deblock,
--rdoq / --no-rdoq, --pu-depth-intra / -inter, --subme, --fast-residual-cost, --no-bipred, --no-tmvp ..., picture sizes that cut CTUs.  Whatever the options, the bitstream must
be the reference encoder's byte for byte: either the binding takes the pictures (and the oracle-served pass must then be the encoder's search) or it has to refuse them
and fall through to kvz_search_lcu -- an eligibility rule that admits a configuration the pass does not reproduce shows up as a different file.
usage: tools/fuzz_binding.py [rounds] [seed]"""
import hashlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inter_common as ic

REF = os.path.join(ROOT, "oracle", "_ref")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def encode(binary, yuv, res, out, opts, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([os.path.join(REF, binary), "-i", yuv, "--input-res", res, "-o", out] + opts, env=e, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return "rc%d" % r.returncode, r.stderr[-300:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), ""


bad = used_intra = used_inter = 0
for r in range(rounds):
    w, h = int(rng.choice([64, 72, 136, 200, 264])), int(rng.choice([64, 88, 136, 200]))
    n = int(rng.integers(2, 7))
    near = rng.integers(0, 2) == 1  # half of the rounds stay close to what the inter pass covers (one or two perturbations), so that B pictures do reach it
    opts = ["--preset", str(rng.choice(["ultrafast", "superfast", "veryfast", "faster"] if near else ["ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow"])),
            "-q", str(int(rng.integers(8, 46)))]
    if not near and rng.integers(0, 2):
        opts += ["-p", "1"]
    else:
        opts += ["--gop", str(rng.choice(["lp-g4d3t1", "lp-g8d4t1", "lp-g2d2t1"]))]
        if rng.integers(0, 4) == 0:
            opts += ["--period", str(int(rng.integers(2, 6)))]
    opts += ["--owf", "0" if near else str(int(rng.choice([0, 0, 1, 2])))]
    scale = 0.25 if near else 1.0
    for o, pr in (("--no-wpp", 0.2), ("--no-deblock", 0.2), ("--no-bipred", 0.1), ("--no-tmvp", 0.1), ("--no-early-skip", 0.1), ("--signhide", 0.1), ("--mv-rdo", 0.05),
                  ("--smp", 0.05), ("--transform-skip", 0.05), ("--full-intra-search", 0.05), ("--intra-rdo-et", 0.05), ("--lossless", 0.02), ("--cqmfile-free", 0.0)):
        if rng.random() < pr * scale:
            opts.append(o)
    for o, vals, pr in (("--sao", ["off", "full", "edge", "band"], 0.3), ("--rdoq", None, 0.15), ("--no-rdoq", None, 0.15), ("--pu-depth-intra", ["1-3", "2-3", "1-4", "2-4", "0-3"], 0.2),
                        ("--pu-depth-inter", ["1-2", "1-3", "0-3", "2-3"], 0.2), ("--subme", ["0", "1", "2", "3", "4"], 0.2), ("--fast-residual-cost", ["0", "20", "28", "40"], 0.2),
                        ("--tiles", ["2x1", "1x2", "2x2"], 0.1), ("--ref", ["1", "2"], 0.1), ("--max-merge", ["2", "5"], 0.1), ("--me", ["hexbs", "tz", "dia"], 0.1),
                        ("--me-early-termination", ["off", "on", "sensitive"], 0.1), ("--rd", ["0", "1", "2"], 0.1), ("--tr-depth-intra", ["0", "1"], 0.05),
                        ("--cu-split-termination", ["zero", "off"], 0.05), ("--slices", ["tiles", "wpp"], 0.03), ("--vaq", ["5"], 0.03), ("--bitrate", ["300000"], 0.02)):
        if rng.random() < pr * scale:
            opts += [o] if vals is None else [o, str(rng.choice(vals))]
    with tempfile.TemporaryDirectory() as d:
        yuv = os.path.join(d, "in.yuv")
        open(yuv, "wb").write(b"".join(f.tobytes() for f in ic.clip(w, h, n, int(rng.integers(1, 1 << 30)), float(rng.uniform(0, 3)), (float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6))))))
        common = opts + ["--threads", "4"]
        want, err = encode("kvazaar_ref", yuv, "%dx%d" % (w, h), os.path.join(d, "ref.hevc"), common)
        if want.startswith("rc"):
            print("round %d: %s: the reference rejects these options (%s)" % (r, " ".join(opts), err.strip().splitlines()[-1] if err.strip() else ""), flush=True)
            continue
        env = {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": os.path.join(d, "ti"), "KVZ_HIP_INTER_TRACE": os.path.join(d, "tb")}
        entropy = int(rng.integers(0, 3) == 0)  # a third of the rounds also hand the slice data of the batched pictures to the (oracle-served) device coder
        if entropy:
            env["KVZ_HIP_BATCH_ENTROPY"] = "1"
        got, err = encode("kvazaar_hipsim", yuv, "%dx%d" % (w, h), os.path.join(d, "sim.hevc"), common, env)
        ti = int(open(os.path.join(d, "ti")).read().split()[0]) if os.path.exists(os.path.join(d, "ti")) else 0
        tb = int(open(os.path.join(d, "tb")).read().split()[0]) if os.path.exists(os.path.join(d, "tb")) else 0
    used_intra += ti > 0; used_inter += tb > 0
    ok = got == want
    print("round %d: %dx%d x %d %s%s: device pictures %d I / %d B -> %s" % (r, w, h, n, " ".join(opts), " [device entropy coding]" if entropy else "", ti, tb, "equal" if ok else "DIFFERENT %s" % err), flush=True)
    bad += not ok
print("%d of %d rounds differ (the binding took I pictures in %d rounds, B pictures in %d)" % (bad, rounds, used_intra, used_inter))
sys.exit(1 if bad else 0)
