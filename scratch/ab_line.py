"""one line per bench JSON: python scratch/ab_line.py <tag> <file.json>"""
import json, sys
d = json.load(open(sys.argv[2]))
its = {}
for s in d.get("per_step", []):
    its.setdefault(s["iteration"], []).append(s["search_ms"])
su = d.get("setup", {})
print("%-10s ms/step %.4f  search/iter %s  normals %s build %s order %s" % (
    sys.argv[1], d["ms_per_step"], [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:8],
    su.get("normals_kernel_ms"), su.get("index_build_ms"), su.get("source_order_ms")))
