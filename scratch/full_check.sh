#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for v in default base; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align > $OUT/$v.json 2> $OUT/$v.err
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align --reciprocal > $OUT/${v}_recip.json 2> $OUT/${v}_recip.err
  python - "$v" "$OUT/$v.json" "$OUT/${v}_recip.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open(sys.argv[2])); s = d["setup"]; r = json.load(open(sys.argv[3]))
print("%-10s ms/step %.4f build %.3f first %.3f source_order %.3f normals %.3f | reciprocal ms/step %.4f" % (sys.argv[1], d["ms_per_step"], s["index_build_ms"], s["index_build_first_ms"], s["source_order_ms"], s["normals_kernel_ms"], r["ms_per_step"]))
PY
done
