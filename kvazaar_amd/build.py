"""Builds libkvz_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.environ.get("KVZ_HIP_LIB") or os.path.join(LIB_DIR, "libkvz_hip.so")  # KVZ_HIP_LIB: developer override (kernel variants)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["kvz_hip.hip"]
# -ffp-contract=off: pixel_var / cost arithmetic must not be contracted into FMAs (bit-exact double results)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    inc = os.path.join(os.path.dirname(PKG), "include")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(p) > t for p in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source into kvazaar_amd/lib/libkvz_hip.so.  Returns the library path."""
    if os.environ.get("KVZ_HIP_LIB") or (not force and not _stale()):
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


TOOLS = {"valu_issue_bench": os.path.join(os.path.dirname(PKG), "tools", "valu_issue_bench.hip")}


def build_tools(force=False, verbose=False):
    """Measurement helpers that run on the GPU box next to the library (kvazaar_amd/lib/<name>): tools/valu_issue_bench.hip."""
    os.makedirs(LIB_DIR, exist_ok=True)
    for name, src in TOOLS.items():
        out = os.path.join(LIB_DIR, name)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
            continue
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", out, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
