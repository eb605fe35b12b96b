"""Aggregates rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ga::", "")
            if "at::" in name or "elementwise" in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for fn in glob.glob(os.path.join(root, "p1", "**", "*kernel_trace.csv"), recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ga::", "")
            if "at::" in name or "elementwise" in name:
                continue
            acc[name]["duration_us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                if k in row:
                    acc[name][k] = [float(row[k])]
for name, ctrs in sorted(acc.items()):
    print("==", name)
    for c, v in sorted(ctrs.items()):
        print("   %-32s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
