#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_gpu.sh into profiles/pmc_traffic.json:
HBM bytes per launch of the ICP kernels, stamped with the commit and the command that produced them.
Usage: scripts/make_pmc_traffic.py <prof dir> <commit> "<command>"

Units and corrections (MI355X_MICROARCH.md, section HBM): both counters are kilobytes; on gfx950 FETCH_SIZE tallies a
128-byte request of a wide coalesced read as 64 bytes, so the read side is doubled (an upper bound for this
gather / LDS-DMA mix); WRITE_SIZE is taken as reported."""
import collections
import csv
import datetime
import glob
import json
import os
import sys

d, commit, cmd = sys.argv[1], sys.argv[2], sys.argv[3]


def kernel_sources_sha():
    """sha256 over the sources of the search kernel (bench.py computes the same and drops a stale file's figures)"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pcl_amd", "csrc")
    h = hashlib.sha256()
    for f in ("search.hip", "traverse.hpp", "pclhip_wave_reduce.hpp", "standoff.hpp", "pclhip_internal.hpp", "index_build.hip"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
    acc = collections.defaultdict(float)
    name_of = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        acc[key] += float(r["Counter_Value"])
        name_of[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (disp, counter), v in acc.items():
        k = name_of[disp]
        short = "icp_search" if ("search_kernel" in k or "search_dual_kernel" in k) and "icp_" in k else \
            "icp_accumulate" if "icp_accumulate_kernel" in k else "normals" if "normals_kernel" in k else None
        if short:
            per[short][counter].append(v)
out = {"commit": commit, "kernel_sources_sha": kernel_sources_sha(), "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "command": cmd,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), scripts/profile_gpu.sh",
       "correction": "KB -> bytes; gfx950: FETCH_SIZE doubled (64 B tallied per 128-B request), WRITE_SIZE as reported"}
for short, c in per.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fk = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        wk = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        out[short + "_FETCH_SIZE_KB_per_launch"] = round(fk, 1)
        out[short + "_WRITE_SIZE_KB_per_launch"] = round(wk, 1)
        out[short + "_launches"] = len(c["FETCH_SIZE"])
        out[short + "_bytes_per_launch"] = int(2 * fk * 1024 + wk * 1024)
json.dump(out, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
