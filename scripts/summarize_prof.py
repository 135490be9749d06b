#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) as text: top kernels by time, and per
kernel the average counter value per dispatch (summed over the hardware instances rocprof reports)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
KEYS = ("icp_search", "icp_cold_search", "icp_accumulate", "icp_iterate", "normals_kernel", "knn_reg", "knn_heap", "vg_", "kd_", "radix", "finalize")
for f in sorted(glob.glob(os.path.join(d, "*kernel_stats.csv"))):
    print("# kernel stats (%s)" % os.path.basename(f))
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-72s calls %5s avg_us %10.1f total_ms %9.2f %6s%%" %
              (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
               r["Percentage"][:6]))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = next((k for k in KEYS if k in name), None)
        if short is None:
            continue
        if "icp_iterate" in name:
            short = "icp_iterate<%s>" % name.split("icp_iterate_kernel<")[1][0]
        per[short][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[short][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("# counters (%s): mean per dispatch; [last] = last dispatch (steady state)" % os.path.basename(f))
    for short, disp in per.items():
        ids = sorted(disp, key=int)
        names = sorted(disp[ids[0]])
        mean = {c: sum(disp[i][c] for i in ids) / len(ids) for c in names}
        last = disp[ids[-1]]
        print("  %-18s n=%3d mean_us %9.1f last_us %9.1f" % (short, len(ids), sum(dur[short].values()) / len(ids),
                                                             dur[short][ids[-1]]))
        for c in names:
            print("      %-24s mean %14.5g   last %14.5g" % (c, mean[c], last[c]))
