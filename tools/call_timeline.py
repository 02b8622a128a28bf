#!/usr/bin/env python3
"""Kernel timeline of the LAST query_gpu() call found in a rocprofv3 kernel-trace database (sqlite):

    rocprofv3 --kernel-trace -d OUT -o kt -- tests/harness/time_results_f32 -n 10000000 -d 128 -k 10 -y 10000 -o 6 -S 12345 -C 16
    python tools/call_timeline.py OUT

One line per kernel from the call's first copy_in_kernel to its last kernel: start (us since the first), duration, stream."""
import glob
import sqlite3
import sys

f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
# the last synchronous call = the last run of kernels that starts with a copy_in_kernel and contains exactly one stage-2 kernel
starts = [i for i, r in enumerate(rows) if r[0].startswith("copy_in_kernel") and (i == 0 or not rows[i - 1][0].startswith("copy_in_kernel")
          and not rows[i - 1][0].startswith("void codes"))]
want = int(sys.argv[2]) if len(sys.argv) > 2 else -1
i0 = starts[want]
t0 = rows[i0][1]
for r in rows[i0:]:
    print("%9.1f us  +%8.1f us  stream %-3s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].split("(")[0][:70]))
    if "stage2" in r[0]:
        print("call span on the GPU: %.1f us" % ((r[2] - t0) / 1e3))
        break
