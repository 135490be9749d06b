#!/bin/bash
# bench A/B of the build numbers: default vs the listed variants, two rounds each (interleaved)
OUT=gpurun_out/$1; mkdir -p $OUT; shift
for rep in 1 2; do
for v in default "$@"; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align > $OUT/$v.json 2> $OUT/$v.err
  python - "$v" "$OUT/$v.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open(sys.argv[2])); s = d["setup"]
print("%-10s ms/step %.4f build %.3f first %.3f source_order %.3f" % (sys.argv[1], d["ms_per_step"], s["index_build_ms"], s["index_build_first_ms"], s["source_order_ms"]))
PY
done
done
