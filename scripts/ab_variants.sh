#!/bin/bash
# A/B of compile-time variants (scripts/build_variant.sh) over the bench: scripts/ab_variants.sh <tag> "<variants>" [bench args]
#   -> gpurun_out/<tag>/ab.log + one summary line per build (ms/step, search ms per iteration of an alignment)
set -u
TAG=$1; VARS=$2; shift; shift
ARGS=${@:-"--no-cpu-baseline --no-host-align --no-families"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in default $VARS; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  [ -f $L ] || continue
  PCLHIP_LIB=$L timeout 300 python bench.py $ARGS > $OUT/$v.json 2> $OUT/$v.err
  python - "$v" "$OUT/$v.json" <<'PY' | tee -a $OUT/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    its = {}
    for s in d.get("per_step", []):
        its.setdefault(s["iteration"], []).append(s["search_ms"])
    print("%-16s ms/step %.4f  search per iteration %s  normals %.3f build %.3f" % (
        sys.argv[1], d["ms_per_step"], [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:6],
        d.get("setup", {}).get("normals_kernel_ms", 0), d.get("setup", {}).get("index_build_ms", 0)))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
