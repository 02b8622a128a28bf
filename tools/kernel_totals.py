#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace CSV (name, launches, total ms, average us), largest first.

    python tools/kernel_totals.py <kernel_trace.csv> [top]
"""
import collections
import csv
import sys

per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    per[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(sum(v) for v in per.values())
print("all kernels: %.1f ms" % (tot / 1e3))
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-72s %6d launches %12.2f ms  avg %12.1f us" % (name, len(v), sum(v) / 1e3, sum(v) / len(v)))
