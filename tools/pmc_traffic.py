"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2) over
tools/eval_probe.py into per-launch HBM traffic of the trunk conv kernel, with the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half the bytes of wide coalesced reads (x2), units are KiB.
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel-name-substring] > profiles/r01_conv_traffic.json
"""
import csv
import json
import sys


def avg(path, counter, needle):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


needle = sys.argv[3] if len(sys.argv) > 3 else "k_resblock"
f, nf = avg(sys.argv[1], "FETCH_SIZE", needle)
w, nw = avg(sys.argv[2], "WRITE_SIZE", needle)
out = dict(kernel=needle, kernel_tag=needle, launches=[nf, nw], fetch_size_kib=f, write_size_kib=w, fetch_bytes=f * 1024 * 2, write_bytes=w * 1024,
           bytes_per_launch=f * 1024 * 2 + w * 1024,
           note="FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), KiB -> bytes; Infinity-Cache hits are counted, not excluded")
print(json.dumps(out, indent=1))
