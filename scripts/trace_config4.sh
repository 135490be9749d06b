#!/bin/bash
# HIP API + kernel trace of the config-4 pipeline (no counters): where the host time of a pass goes.
# usage (gpurun): bash scripts/trace_config4.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-c4trace}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rp_c4
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -d /tmp/rp_c4 -o c4 --output-format csv -- python $REPO/bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/run.log 2>&1
echo rc=$?
find /tmp/rp_c4 -name "*stats*.csv" | while read f; do cp "$f" $OUT/$(basename "$f"); done
for f in $OUT/*hip_api_stats.csv $OUT/*kernel_stats.csv; do echo "== $f"; head -25 "$f" | cut -c1-160; done
tail -2 $OUT/run.log | cut -c1-600
