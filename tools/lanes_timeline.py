#!/usr/bin/env python3
"""Kernel timeline of the last steps of a rocprofv3 --kernel-trace run, with the queue/stream of every kernel, to see
whether batches on different HIP streams really overlap.

    python tools/lanes_timeline.py <kernel_trace.csv> [n_last_kernels]
"""
import csv
import sys

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f -> %9.1f us  dur %8.1f  q%-3s s%-3s %-40s grid %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"],
          r["Stream_Id"], r["Kernel_Name"].split("(")[0][:40], r["Grid_Size_X"]))
