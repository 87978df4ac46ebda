#!/usr/bin/env python3
"""Static VALU opcode mix of the CTU kernel (the QP < 28 instantiation) from the built library's gfx950 code object, classed by the
issue cost tools/valu_issue_bench.hip measured: `fast` = plain 32-bit VOP1/VOP2 (v_mov, v_add/sub_u32, v_and/or/xor, v_lshrrev/ashrrev, ...
~1.0-1.1 ns of SIMD time per wave64 instruction at saturation), `slow` = everything else (VOP3, packed, SDWA, DPP, conversions, f64, ...
~1.78 ns).  Prints JSON.  usage: tools/valu_mix.py [libkvz_hip.so]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "kvazaar_amd", "lib", "libkvz_hip.so")
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, "lib.so")
    os.symlink(lib, tmp)
    subprocess.run([objdump, "--offloading", tmp], cwd=d, capture_output=True)
    m = None
    for co in sorted(f for f in os.listdir(d) if "gfx950" in f):  # one code object per translation unit (kvazaar_amd/build.py): the kernel is in one of them
        dis = subprocess.run([objdump, "-d", os.path.join(d, co)], capture_output=True, text=True).stdout
        m = re.search(r"\n[0-9a-f]+ <_ZN3kvz23intra_ctu_ticket_kernelILb0ELb0ELb0E[^>]*>:\n(.*?)(?:\n[0-9a-f]+ <[^>]*>:\n|\Z)", dis, re.S)  # up to the next symbol
        if m:
            break
ops = collections.Counter(line.split()[0] for line in m.group(1).splitlines() if line.strip() and line.split()[0][:2] in ("v_", "s_", "ds", "gl", "bu", "fl", "sc"))
# measured ~1.0-1.1 ns class (tools/valu_issue_bench.hip): e32 encodings of the plain integer / logic / move / f32-fma ops
FAST = re.compile(r"^v_(mov_b32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|lshrrev_b32|ashrrev_i32|not_b32|fma_f32|add_f32|mul_f32|max_u32|min_u32|max_i32|min_i32)_e32$")
valu = {k: v for k, v in ops.items() if k.startswith("v_") and not k.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mfma", "v_accvgpr")) or k.startswith(("v_readlane", "v_readfirstlane", "v_writelane"))}
fast = sum(v for k, v in valu.items() if FAST.match(k))
total = sum(valu.values())
print(json.dumps({"kernel": "intra_ctu_ticket_kernel<false>", "valu_instructions_static": total, "fast_class_share": fast / total,
                  "top": collections.Counter(valu).most_common(25)}))
