#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT; shift
timeout 600 python -m pytest tests -m gpu -x -q -k "voxel or pipeline" > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log | tail -2
for rep in 1 2; do
for v in default "$@"; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  PCLHIP_LIB=$L timeout 300 python bench.py --config 4 --no-cpu-baseline --no-host-align > $OUT/${v}_c4.json 2> $OUT/${v}_c4.err
  python - "$v" "$OUT/${v}_c4.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
r = json.load(open(sys.argv[2]))
print("%-10s config 4 ms/step %.4f %s" % (sys.argv[1], r["ms_per_step"], r["stages_ms"]))
PY
done
done
bash scratch/prof_vg.sh $(basename $OUT) > /dev/null 2>&1
