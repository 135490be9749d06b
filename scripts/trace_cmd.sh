#!/bin/bash
# scripts/trace_cmd.sh <outdir> -- <command...> : rocprofv3 --kernel-trace --stats of a command on the GPU box; the per-kernel
# summary (calls, average, total) lands in <outdir>/kernel_stats.txt
set -u
OUT=$1; shift; shift
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/rp_trace
( cd /tmp && cd $REPO && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_trace -o trace --output-format csv -- "$@" ) > $OUT/trace.log 2>&1
f=$(find /tmp/rp_trace -name "*kernel_stats.csv" | head -1)
python - "$f" > $OUT/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-86s calls %5s avg_us %10.1f total_ms %9.2f %6s%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  float(r["TotalDurationNs"]) / 1e6, r["Percentage"][:6]))
PY
tail -3 $OUT/trace.log
