#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / PMC counter collection) into small text summaries
for profiles/.  Usage: summarize_prof.py <dir-with-csv> <out.txt> [filter-substring]"""
import collections
import csv
import glob
import os
import sys


def short(name, n=110):
    return name if len(name) <= n else name[:n] + "..."


def main():
    src, out = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    lines = []
    for f in sorted(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)):
        lines.append("# %s" % os.path.relpath(f, src))
        lines.append("%-112s %6s %14s %14s %8s %12s %12s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
        for r in csv.DictReader(open(f)):
            if flt and flt not in r["Name"]:
                continue
            lines.append("%-112s %6s %14s %14.1f %8s %12s %12s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                                  float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
    for f in sorted(glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            if flt and flt not in r["Kernel_Name"]:
                continue
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
        lines.append("# %s" % os.path.relpath(f, src))
        lines.append("%-112s %-22s %6s %18s   grid/wg/lds/vgpr/sgpr" % ("kernel", "counter", "calls", "avg_per_dispatch"))
        for k, v in sorted(acc.items()):
            lines.append("%-112s %-22s %6d %18.3f   %s" % (k[0], k[1], len(v), sum(v) / len(v), "/".join(meta[k])))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
