#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT; shift
for v in default "$@"; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align > $OUT/$v.json 2> $OUT/$v.err
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align --reciprocal > $OUT/${v}_recip.json 2> $OUT/${v}_recip.err
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align --rejectors median,trimmed > $OUT/${v}_rej.json 2> $OUT/${v}_rej.err
  PCLHIP_LIB=$L timeout 300 python scratch/knn_probe.py 10000000 1 8 > $OUT/${v}_knn.log 2>&1
  python - "$v" "$OUT/$v.json" "$OUT/${v}_recip.json" "$OUT/${v}_rej.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open(sys.argv[2])); s = d["setup"]; r = json.load(open(sys.argv[3])); j = json.load(open(sys.argv[4]))
its = {}
for x in d.get("per_step", []):
    its.setdefault(x["iteration"], []).append(x["search_ms"])
print("%-8s ms/step %.4f %s build %.3f source_order %.3f | reciprocal %.4f | rejectors %.4f" % (sys.argv[1], d["ms_per_step"], [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:5], s["index_build_ms"], s["source_order_ms"], r["ms_per_step"], j["ms_per_step"]))
PY
  grep -v "^/opt\|^$" $OUT/${v}_knn.log | tail -4 | tee -a $OUT/ab.log
done
