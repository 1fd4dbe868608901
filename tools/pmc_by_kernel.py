"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/profile_round.sh) with the gfx950 corrections
of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE x 2 for wide coalesced reads, KiB units), and the kernels' average durations
from the kernel trace of the same (profiled) runs.
    python tools/pmc_by_kernel.py <dir with pmc_f/ and pmc_w/> > profiles/<tag>_hbm_traffic_by_kernel.json"""
import csv
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def counters(sub, name):
    acc = defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, sub, "run_counter_collection.csv"))):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(sub):
    acc = defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, sub, "run_kernel_trace.csv"))):
        acc[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return acc


f, w, t = counters("pmc_f", "FETCH_SIZE"), counters("pmc_w", "WRITE_SIZE"), durations("pmc_f")
out = []
for k in sorted(f, key=lambda k: -sum(t[k])):
    if len(f[k]) < 10:
        continue
    fb = sum(f[k]) / len(f[k]) * 1024 * 2
    wb = sum(w[k]) / len(w[k]) * 1024 if k in w else 0.0
    us = sum(t[k]) / len(t[k]) / 1e3
    out.append(dict(kernel=k.split("(")[0], launches=len(f[k]), fetch_bytes=fb, write_bytes=wb, avg_us_profiled=us, gb_per_s=(fb + wb) / us / 1e3))
print(json.dumps(out, indent=1))
