#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the two counter passes of tools/profile_round.sh.
usage: tools/make_pmc_traffic_json.py <tag> [width height frames qp]   (reads gpurun_out/<tag>_pmc_{f,w}/*counter_collection.csv)"""
import csv, glob, json, sys

def total(path, counter, kernel):
    s, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
            s += float(row["Counter_Value"]); n += 1
    return s, n

def main():
    tag = sys.argv[1]
    w, h, frames, qp = (int(v) for v in (sys.argv[2:6] if len(sys.argv) >= 6 else (1920, 1080, 1536, 22)))
    kernel = "intra_ctu_ticket_kernel"
    f, nf = total(glob.glob(f"gpurun_out/{tag}_pmc_f/*counter_collection.csv")[0], "FETCH_SIZE", kernel)
    wr, nw = total(glob.glob(f"gpurun_out/{tag}_pmc_w/*counter_collection.csv")[0], "WRITE_SIZE", kernel)
    out = {
        "workload": {"width": w, "height": h, "frames": frames, "schedule": "ticket", "kernel": kernel, "qp": qp},
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), python bench.py --steps 1 --warmup 1; "
                  "unit of the counters = KB (x1024 bytes). MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x; the accesses of this "
                  "kernel are narrow (u8/i16 strided), for which the counter is uncalibrated -- raw value reported.",
        "fetch_size_kb": f, "fetch_size_kb_launches": nf, "write_size_kb": wr, "write_size_kb_launches": nw,
        "bytes_per_launch": (f / nf + wr / nw) * 1024.0,
    }
    json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out))

if __name__ == "__main__":
    main()
