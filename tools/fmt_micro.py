#!/usr/bin/env python3
"""Pretty-prints the JSON lines of bench_kernels.py (stdin)."""
import json
import sys
for l in sys.stdin:
    try:
        d = json.loads(l)
    except ValueError:
        print(l.rstrip())
        continue
    if "kernel" in d:
        m = f"  mfma {d['mfma_algorithmic_frac']:.4f} (issued {d['mfma_issued_frac']:.4f})" if "mfma_algorithmic_frac" in d else ""
        print(f"{d['kernel']:18s} {d['blocks']:9d} blk {d['ms']:8.3f} ms {d['achieved_GBps']:8.1f} GB/s  frac {d['frac']:.3f}{m}")
