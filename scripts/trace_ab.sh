#!/bin/bash
# kernel-trace A/B of library variants over the bench: average duration of the named kernels.
# usage: scripts/trace_ab.sh <tag> "<variants>" "<kernel substrings>"
TAG=$1; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for v in $2; do
  L=$REPO/pcl_amd/libpclhip.so; [ $v != default ] && L=$REPO/pcl_amd/variants/libpclhip_$v.so
  rm -rf /tmp/rp_$v
  PCLHIP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_$v -o t --output-format csv -- python $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-host-align --no-families > $OUT/$v.log 2>&1
  f=$(find /tmp/rp_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  cp "$f" $OUT/${v}_kernel_stats.csv
  python3 - "$f" $3 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for k in sys.argv[2:]:
    for r in rows:
        if k in r["Name"]:
            print("  %-44s calls %5s avg_us %9.2f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done | tee $OUT/summary.txt
