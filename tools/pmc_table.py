#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel launch geometry.

    python tools/pmc_table.py <substring of kernel name> <counter_collection.csv> [more.csv ...]

Prints one markdown table: a row per (kernel, grid size), a column per counter found in the CSVs (values averaged over
the launches of that geometry).  Kernel names are cut at the first '(' and to 48 characters.
"""
import collections
import csv
import sys


def main():
    pat, paths = sys.argv[1], sys.argv[2:]
    csv.field_size_limit(1 << 30)
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    counters = []
    for p in paths:
        for r in csv.DictReader(open(p)):
            name = r["Kernel_Name"].split("(")[0]
            if pat not in name:
                continue
            key = (name[:48], int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
            c = r["Counter_Name"]
            if c not in counters:
                counters.append(c)
            vals[key][c].append(float(r["Counter_Value"]))
    print("| kernel | grid | wg | vgpr | lds | launches | " + " | ".join(counters) + " |")
    print("|" + "---|" * (6 + len(counters)))
    for key in sorted(vals, key=lambda k: (k[0], -k[1])):
        v = vals[key]
        n = max(len(x) for x in v.values())
        cells = ["%.4g" % (sum(v[c]) / len(v[c])) if c in v else "-" for c in counters]
        print("| %s | %d | %d | %d | %d | %d | %s |" % (key[0], key[1], key[2], key[3], key[4], n, " | ".join(cells)))


if __name__ == "__main__":
    main()
