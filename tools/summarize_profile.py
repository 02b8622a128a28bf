#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel-trace/--stats CSVs and the separate --pmc FETCH_SIZE / WRITE_SIZE passes,
collected as DESIGN.md section 6 describes) into a small markdown summary under profiles/.

    python tools/summarize_profile.py <kernel_trace.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> \
           <query_grid_size> <out.md> [title]

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a 16-B-per-lane
coalesced read stream -> doubled here; WRITE_SIZE is exact.  Both counters are in KiB.
"""
import collections
import csv
import sys


def main():
    kt, fetch, write, qgrid, out = sys.argv[1:6]
    title = sys.argv[6] if len(sys.argv) > 6 else "rocprofv3 summary"
    rows = list(csv.DictReader(open(kt)))
    per = collections.defaultdict(list)
    for r in rows:
        per[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["# %s" % title, "", "## per-kernel durations (rocprofv3 --kernel-trace), grouped by launch geometry", "",
             "| kernel | grid | wg | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|---|---|"]
    for (name, g, w), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| %s | %s | %s | %d | %.1f | %.1f | %.1f | %.2f |" % (name, g, w, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))

    def pmc(path, counter):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                vals[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
        return vals
    f, w = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
    lines += ["", "## HBM traffic per launch (separate --pmc passes; KiB counters; FETCH_SIZE doubled per the gfx950 note)", "",
              "| kernel | grid | launches | FETCH_SIZE raw KiB | read GB (x2 corrected) | WRITE_SIZE KiB | written GB |", "|---|---|---|---|---|---|---|"]
    for key in sorted(f, key=lambda k: -sum(f[k]) / len(f[k])):
        fv = sum(f[key]) / len(f[key])
        wv = sum(w.get(key, [0])) / max(len(w.get(key, [0])), 1)
        mark = " **(query launch)**" if key[1] == qgrid and "stage1" in key[0] else ""
        lines.append("| %s%s | %s | %d | %.0f | %.3f | %.0f | %.3f |" % (key[0], mark, key[1], len(f[key]), fv, fv * 1024 * 2 / 1e9, wv, wv * 1024 / 1e9))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
