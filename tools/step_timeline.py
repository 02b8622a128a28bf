#!/usr/bin/env python3
"""Print the kernel timeline of the last two query steps from a rocprofv3 kernel_trace.csv (bench.py run)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qgrid = sys.argv[2] if len(sys.argv) > 2 else "2560000"
idx = [i for i, r in enumerate(rows) if "stage1_select" in r["Kernel_Name"] and r["Grid_Size_X"] == qgrid]
i0 = idx[-2] - 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0: idx[-1] + 8]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  gap %6.1f  dur %8.1f us  %-44s grid %-9s wg %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
          r["Kernel_Name"].split("(")[0][:44], r["Grid_Size_X"], r["Workgroup_Size_X"]))
    prev_end = e
