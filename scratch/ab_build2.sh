#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
for v in default base; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align > $OUT/$v.json 2> $OUT/$v.err
  PCLHIP_LIB=$L timeout 300 python bench.py --config 4 --no-cpu-baseline --no-host-align > $OUT/${v}_c4.json 2> $OUT/${v}_c4.err
  python - "$v" "$OUT/$v.json" "$OUT/${v}_c4.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open(sys.argv[2])); s = d["setup"]; r = json.load(open(sys.argv[3]))
print("%-10s ms/step %.4f build %.3f first %.3f source_order %.3f | config 4 ms/step %.4f %s" % (sys.argv[1], d["ms_per_step"], s["index_build_ms"], s["index_build_first_ms"], s["source_order_ms"], r["ms_per_step"], json.dumps(r.get("pipeline") or r.get("setup"))[:400]))
PY
done
bash scratch/prof_build.sh $1 > /dev/null 2>&1
