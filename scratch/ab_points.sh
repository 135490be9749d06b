#!/bin/bash
# bench.py at another cloud size: default vs variants (one run each)
OUT=gpurun_out/$1; N=$2; shift; shift; mkdir -p $OUT
for v in default "$@"; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  PCLHIP_LIB=$L timeout 200 python bench.py --points $N --no-cpu-baseline --no-host-align > $OUT/$v.json 2> $OUT/$v.err
  python - "$v" "$OUT/$v.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open(sys.argv[2])); s = d["setup"]
its = {}
for x in d.get("per_step", []):
    its.setdefault(x["iteration"], []).append(x["search_ms"])
print("%-8s ms/step %.4f %s build %.3f source_order %.3f normals %.3f" % (sys.argv[1], d["ms_per_step"], [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:5], s["index_build_ms"], s["source_order_ms"], s["normals_kernel_ms"]))
PY
done
