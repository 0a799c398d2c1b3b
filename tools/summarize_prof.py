#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / kernel trace / PMC counter collection) into small text summaries
for profiles/.  Usage: summarize_prof.py <dir-with-csv> <out.txt> [filter-substring]

Besides rocprofv3's own per-kernel statistics (which average EVERY launch, the cold ones of a fresh process included)
the kernel trace yields a steady-state figure per kernel: the average over the launches after the first STEADY_SKIP
(40: a cold process runs its first ~30 launches at ramping clocks, profiles/r02_cfft_f32_4096_launch_series.txt),
or over the last half when a kernel has fewer launches than that.  bench.py's roofline.avg_launch_ms is measured on
the timed steps after preheat + warm-up, i.e. on the same steady state."""
import collections
import csv
import glob
import os
import sys


STEADY_SKIP = 40


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
    for f in sorted(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if flt and flt not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        lines.append("# %s -- steady state: launches after the first %d (or the last half of fewer)" % (os.path.relpath(f, src), STEADY_SKIP))
        lines.append("%-112s %6s %8s %14s %14s %12s %12s" % ("kernel", "calls", "counted", "steady_avg_ns", "steady_med_ns", "min_ns", "max_ns"))
        for name, rows in per.items():
            rows.sort()
            d = [x[1] for x in rows]
            tail = d[STEADY_SKIP:] if len(d) >= 2 * STEADY_SKIP else d[len(d) // 2:]
            if not tail:
                continue
            srt = sorted(tail)
            lines.append("%-112s %6d %8d %14.1f %14.1f %12d %12d" % (short(name), len(d), len(tail), sum(tail) / len(tail), srt[len(srt) // 2], srt[0], srt[-1]))
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
