"""MFMA utilisation and LDS bank conflicts per kernel from the two extra rocprofv3 --pmc passes of tools/profile_round.sh
(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; SQ_LDS_BANK_CONFLICT + SQ_LDS_IDX_ACTIVE).  SQ_VALU_MFMA_BUSY_CYCLES sums the cycles the
matrix pipes of all SIMDs are busy; GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles, so the kernel ran GRBM_GUI_ACTIVE / 8 cycles on
`simds` SIMDs (MI355X_MICROARCH.md: cycle constants, DVFS give-back).
    python tools/pmc_mfma.py <dir with pmc_m/ and pmc_l/> [simds = 1024] > profiles/<tag>_mfma_util.json"""
import csv
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
simds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    path = os.path.join(d, sub, "run_counter_collection.csv")
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


m, l = counters("pmc_m"), counters("pmc_l")
out = []
for k in m:
    c = m[k]
    if len(c.get("GRBM_GUI_ACTIVE", [])) < 10:
        continue
    mean = lambda v: sum(v) / len(v) if v else 0.0
    busy, gui = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(c["GRBM_GUI_ACTIVE"])
    row = dict(kernel=k.split("(")[0], launches=len(c["GRBM_GUI_ACTIVE"]), mfma_busy_cycles=busy, kernel_cycles=gui / 8,
               mfma_util=busy / (gui / 8 * simds) if gui else None)
    if k in l:
        row["lds_bank_conflict_cycles"] = mean(l[k].get("SQ_LDS_BANK_CONFLICT", []))
        row["lds_active_cycles"] = mean(l[k].get("SQ_LDS_IDX_ACTIVE", []))
    out.append(row)
out.sort(key=lambda r: -r["kernel_cycles"] * r["launches"])
print(json.dumps(out, indent=1))
