#!/usr/bin/env python3
"""profiles/<tag>_pmc_sq.json from the committed measurements of a round: SQ counters of the bench workload (tools/pmc_sq.sh log), the
measured SIMD time per wave64 VALU instruction (tools/valu_issue_bench.hip output) and the static opcode mix of the CTU kernel
(tools/valu_mix.py).  usage: tools/make_pmc_sq_json.py <tag> <kernel_ms>   (files profiles/<tag>_{pmc_sq.log,valu_issue.jsonl,valu_mix.json})"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, kernel_ms = sys.argv[1], float(sys.argv[2])
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")
c = {}
for line in open(P("pmc_sq.log")):
    m = re.match(r"(SQ_\w+)\s+([0-9.e+]+) per launch", line)
    if m:
        c[m.group(1)] = float(m.group(2))
ns = {}
for line in open(P("valu_issue.jsonl")):
    d = json.loads(line)
    if "instruction" in d:
        ns[d["instruction"]] = d["k8"]["simd_ns_per_wave_inst"]
mix = json.load(open(P("valu_mix.json")))
fast_ns = sum(ns[k] for k in ("v_mov_b32", "v_and_b32", "v_add_u32", "v_sub_u32", "v_lshrrev_b32")) / 5
slow = [v for k, v in ns.items() if k not in ("v_mov_b32", "v_and_b32", "v_add_u32", "v_sub_u32", "v_lshrrev_b32", "v_fma_f32", "v_cndmask_b32(vcc)")]
slow_ns = sum(slow) / len(slow)
simds, t = 1024, kernel_ms / 1e3
share = mix["fast_class_share"]
frac = lambda per_inst_ns: c["SQ_INSTS_VALU"] * per_inst_ns * 1e-9 / (simds * t)
out = {
    "workload": {"width": 1920, "height": 1080, "frames": 1536, "qp": 22, "schedule": "ticket"},
    "kernel_ms": kernel_ms,
    "insts_valu": c["SQ_INSTS_VALU"], "insts_salu": c["SQ_INSTS_SALU"], "insts_lds": c["SQ_INSTS_LDS"], "insts_mfma": c.get("SQ_INSTS_MFMA"),
    "cycles_per_valu_inst": None,
    "simd_ns_per_valu_inst": {"fast_class": fast_ns, "slow_class": slow_ns, "static_fast_share": share,
                              "source": f"profiles/{tag}_valu_issue.jsonl (k8 = saturated SIMDs, wall clock), profiles/{tag}_valu_mix.json"},
    "valu_issue_frac": frac(share * fast_ns + (1 - share) * slow_ns),
    "valu_issue_frac_range": [frac(fast_ns), frac(slow_ns)],
    "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
    "wave_issue_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "wave_wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "wave_issue_stall_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "waves_per_simd": c["SQ_WAVES"] / simds,
    "note": "valu_issue_frac = SQ_INSTS_VALU x (measured SIMD-ns per wave64 VALU instruction, two cost classes weighted by the kernel's static opcode mix) / (1024 SIMDs x kernel time); "
            "the range is all-fast .. all-slow.  lane_utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64).  wave_* = share of SQ_WAVE_CYCLES (quad-cycles) a resident "
            "wavefront spends issuing / parked on s_waitcnt or a barrier / stalled at issue.",
}
json.dump(out, open(P("pmc_sq.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
