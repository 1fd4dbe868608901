"""rocprofv3 --kernel-trace CSV -> launches grouped by (kernel, workgroups): calls, average / min / max duration.
rocprofv3 --stats averages a kernel over ALL its launches; bench.py runs three configurations in one command, and the same trunk kernel serves a
4096-board (Connect4) and an 8192-board (Gumbel) batch — this table keeps them apart so that a launch shape's average can be held against the
bench line's avg_launch_us.  usage: python tools/kernel_stats_by_grid.py <run_kernel_trace.csv> [min_total_ms] > out.csv"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
acc = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        acc[(r["Kernel_Name"], grid // max(wg, 1), wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Workgroups", "WorkgroupSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for (name, n, wg), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(d) / 1e6 >= min_ms:
        w.writerow([name, n, wg, len(d), sum(d), round(sum(d) / len(d), 1), min(d), max(d)])
