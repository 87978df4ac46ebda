"""No-GPU checks of the boundary: libkvz_hip.so builds (hipcc cross-compiles gfx950), loads, and exports every
symbol include/kvz_hip*.h declares.  No compute call is made here."""
import ctypes
import os
import re

import flatapi


def _declared_symbols():
    inc = os.path.join(flatapi.ROOT, "include")
    names = set()
    for f in sorted(os.listdir(inc)):
        if not f.endswith(".h"):
            continue
        text = open(os.path.join(inc, f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(kvz_hip_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_builds_and_exports_every_declared_symbol():
    import kvazaar_amd
    path = kvazaar_amd.build_library()
    lib = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) > 60
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing


def test_exported_symbols_are_kvz_prefixed():
    """tests/test_external_symbols.sh of the reference: every exported symbol must start with kvz_"""
    import subprocess
    import kvazaar_amd
    out = subprocess.check_output(["nm", "-D", "--defined-only", kvazaar_amd.build_library()], text=True)
    bad = [ln.split()[-1] for ln in out.splitlines()
           if ln.split() and ln.split()[-2] in ("T", "D", "B") and not ln.split()[-1].startswith(("kvz_", "_Z", "__hip", "_fini", "_init"))]
    assert not bad, bad


def test_flat_api_complete():
    """every name of the flat API table exists with the hip prefix"""
    import kvazaar_amd
    lib = ctypes.CDLL(kvazaar_amd.build_library())
    missing = [n for n in flatapi.SIGNATURES if not hasattr(lib, "kvz_hip_" + n)]
    assert not missing, missing
