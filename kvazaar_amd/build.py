"""Builds libkvz_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is nine translation units compiled in parallel: kvz_hip.hip (C ABI, per-call ops, streaming kernels, host side of the batch; with
-DKVZ_CTU_SEPARATE_TUS it only declares the CTU kernels) and kvz_ctu_tu.hip six times, one CTU kernel instantiation each (-DKVZ_CTU_KERNEL_TU=0..5,
csrc/kvz_ctu_kernels.hpp).  Objects are rebuilt when one of the files they include (hipcc -MD) is newer."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.environ.get("KVZ_HIP_LIB") or os.path.join(LIB_DIR, "libkvz_hip.so")  # KVZ_HIP_LIB: developer override (kernel variants)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: pixel_var / cost arithmetic must not be contracted into FMAs (bit-exact double results)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# (object name, source, extra defines)
MFMA_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]  # an internal LLVM option, only a tuning of the streaming transform kernels: dropped where hipcc does not know it (_probe_flags)
UNITS = ([("kvz_hip", "kvz_hip.hip", ["-DKVZ_CTU_SEPARATE_TUS"] + MFMA_VGPR_FORM)]  # (the streaming transform kernels: accumulators in VGPRs, no v_accvgpr moves around the bias pass)
         + [(f"kvz_ctu_tu{k}", "kvz_ctu_tu.hip", [f"-DKVZ_CTU_KERNEL_TU={k}"]) for k in range(6)]
         + [("kvz_inter_tu0", "kvz_inter_tu.hip", ["-DKVZ_ICTU_CABAC=0"]), ("kvz_inter_tu1", "kvz_inter_tu.hip", ["-DKVZ_ICTU_CABAC=1"])])  # the inter CTU pass's two builds (csrc/kvz_inter_kernels.hpp)


def _deps(dfile):
    try:
        txt = open(dfile).read().replace("\\\n", " ")
    except OSError:
        return None
    return [t for t in txt.split(":", 1)[1].split() if t] if ":" in txt else None


def _unit_stale(name):
    obj, dfile = os.path.join(OBJ_DIR, name + ".o"), os.path.join(OBJ_DIR, name + ".d")
    deps = _deps(dfile)
    if not os.path.exists(obj) or deps is None:
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(p)) or os.path.getmtime(p) > t for p in deps) or os.path.getmtime(os.path.abspath(__file__)) > t


def _stale():
    return not os.path.exists(LIB_PATH) or any(_unit_stale(n) or os.path.getmtime(os.path.join(OBJ_DIR, n + ".o")) > os.path.getmtime(LIB_PATH) for n, _, _ in UNITS)


_flag_ok = {}


def _probe_flags(flags):
    """does this hipcc accept `flags`?  (an empty translation unit, host side only; cached per process)"""
    key = tuple(flags)
    if key not in _flag_ok:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            open(src, "w").write("int kvz_probe;\n")
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-c", "-o", os.path.join(d, "probe.o"), src] + list(flags), capture_output=True)
            _flag_ok[key] = r.returncode == 0
    return _flag_ok[key]


def _compile_and_link(obj_dir, lib_path, extra_flags, only_stale, verbose):
    os.makedirs(obj_dir, exist_ok=True)
    mfma_form = _probe_flags(MFMA_VGPR_FORM) and not os.environ.get("KVZ_HIP_NO_MFMA_VGPR_FORM")

    def compile_unit(unit):
        name, src, defs = unit
        if only_stale and not _unit_stale(name):
            return
        if not mfma_form:
            defs = [d for d in defs if d not in MFMA_VGPR_FORM]
        obj = os.path.join(obj_dir, name + ".o")
        cmd = [HIPCC] + FLAGS + defs + list(extra_flags) + ["-c", "-MD", "-MF", os.path.join(obj_dir, name + ".d"), "-o", obj, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_unit, UNITS))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + [os.path.join(obj_dir, n + ".o") for n, _, _ in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


def build_library(force=False, verbose=False):
    """Compile every HIP source into kvazaar_amd/lib/libkvz_hip.so.  Returns the library path."""
    if os.environ.get("KVZ_HIP_LIB") or (not force and not _stale()):
        return LIB_PATH
    return _compile_and_link(OBJ_DIR, LIB_PATH, [], not force, verbose)


def build_variant(name, extra_flags, verbose=False, only_units=None):
    """Developer builds with extra compiler flags (-DKVZ_CTU_PROFILE, occupancy experiments ...): kvazaar_amd/lib/variants/libkvz_hip_<name>.so, always rebuilt.
    only_units: names of the translation units the flags matter for (e.g. ["kvz_ctu_tu5"], the RDOQ kernel) -- only those are compiled, the rest are the default
    build's objects.  Use with KVZ_HIP_LIB=<path> (bench.py, tools/)."""
    vdir = os.path.join(LIB_DIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    lib_path = os.path.join(vdir, f"libkvz_hip_{name}.so")
    if not only_units:
        return _compile_and_link(os.path.join(vdir, "obj_" + name), lib_path, extra_flags, False, verbose)
    build_library()
    obj_dir = os.path.join(vdir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    mfma_form = _probe_flags(MFMA_VGPR_FORM) and not os.environ.get("KVZ_HIP_NO_MFMA_VGPR_FORM")
    objs = []
    for uname, src, defs in UNITS:
        if uname not in only_units:
            objs.append(os.path.join(OBJ_DIR, uname + ".o"))
            continue
        if not mfma_form:
            defs = [d for d in defs if d not in MFMA_VGPR_FORM]
        obj = os.path.join(obj_dir, uname + ".o")
        cmd = [HIPCC] + FLAGS + defs + list(extra_flags) + ["-c", "-o", obj, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


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
