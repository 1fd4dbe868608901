"""Mean value of every counter of a rocprofv3 --pmc pass for the kernels whose name contains <needle>.
    python tools/pmc_dump.py <run_counter_collection.csv> [needle = k_trunk]"""
import csv, collections, sys
needle = sys.argv[2] if len(sys.argv) > 2 else "k_trunk"
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if needle in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"{c:40s} {sum(v) / len(v):16.1f}   ({len(v)} launches)")
