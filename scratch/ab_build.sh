#!/bin/bash
# gpurun: build-path tests on the current library, then the bench's setup numbers for default vs variants
TAG=$1; VARS=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kd_order or order or build or fuzz or nonfinite or nan or source" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for v in default $VARS; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  [ -f $L ] || continue
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align > $OUT/$v.json 2> $OUT/$v.err
  python - "$v" "$OUT/$v.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.load(open(sys.argv[2])); s = d["setup"]
print("%-10s ms/step %.4f build %.3f first %.3f source_order %.3f normals %.3f" % (sys.argv[1], d["ms_per_step"], s["index_build_ms"], s["index_build_first_ms"], s["source_order_ms"], s["normals_kernel_ms"]))
PY
done
