#!/bin/bash
# per-dispatch kernel durations of the LAST index build of scratch/build_only.py
TAG=$1; L=${2:-pcl_amd/libpclhip.so}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
PCLHIP_LIB=$L timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o trace --output-format csv -- python scratch/build_only.py ${N:-10000000} > $OUT/run.log 2>&1
grep "^build" $OUT/run.log
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/dispatches.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'pclhip' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last build starts at the last kd_load_box / kd_count dispatch
starts = [i for i, r in enumerate(rows) if 'kd_load_box' in r['Kernel_Name'] or 'kd_count_kernel' in r['Kernel_Name']]
rows = rows[starts[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
tot = 0
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = r['Kernel_Name'].replace('pclhip::(anonymous namespace)::', '').replace('void ', '')[:34]
    print("%8.1f us @%8.1f  %s" % (d, (int(r['Start_Timestamp']) - t0) / 1e3, name))
print("sum of kernels %.1f us, span %.1f us" % (tot, (int(rows[-1]['End_Timestamp']) - t0) / 1e3))
PY
rm -rf $OUT/prof
