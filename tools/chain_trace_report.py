#!/usr/bin/env python3
"""The kernels' timeline out of a `rocprofv3 --kernel-trace --output-format csv` directory: start, end, duration of every kernel longer than 0.3 ms.
usage: tools/chain_trace_report.py <dir> [last N]"""
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 26
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
big = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 3e5]
print(len(rows), "kernels;", len(big), "longer than 0.3 ms; the last", last)
for r in big[-last:]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %8.1f ms  q%s  %s" % (a / 1e6, b / 1e6, (b - a) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
