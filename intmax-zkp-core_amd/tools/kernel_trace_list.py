#!/usr/bin/env python3
"""List every launch of the kernels whose name contains argv[2] from a rocprofv3 kernel trace (argv[1] = directory holding
*_kernel_trace.csv): launch order, grid, workgroup, duration.  Used to see the per-pass cost of the NTT."""
import csv
import glob
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("| # | kernel | grid | wg | ms |")
print("|---|---|---|---|---|")
for i, r in enumerate(rows):
    g = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r) or r.get("Grid_Size", "")
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("| %d | %s | %s | %s | %.3f |" % (i, r["Kernel_Name"][:60], g, r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), ms))
