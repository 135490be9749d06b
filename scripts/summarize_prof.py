#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a compact text/JSON report."""
import csv, glob, json, os, sys
from collections import defaultdict
d = sys.argv[1]
rep = {}
for f in sorted(glob.glob(os.path.join(d, "*kernel_stats.csv"))):
    rows = list(csv.DictReader(open(f)))
    rep["kernel_stats"] = [{k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:12]]
pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")[:60]
        key = r.get("Counter_Name"); val = float(r.get("Counter_Value", 0) or 0)
        disp = r.get("Dispatch_Id")
        a = pmc[name][key]; a[0] += val; a[1] += 1
# per-dispatch average: counters are reported per dispatch (possibly one row per XCD/SE instance)
out = {}
for name, cs in pmc.items():
    out[name] = {k: {"sum": v[0], "rows": v[1]} for k, v in cs.items()}
rep["pmc"] = out
print(json.dumps(rep, indent=1))
