#!/bin/bash
# Per-iteration PMC passes of scratch/iter_probe.py (6 ICP iterations from the cold start, twice).
# Usage: scripts/profile_iter.sh <tag>   -> gpurun_out/prof_<tag>/per_iter.txt
set -u
TAG=${1:-iter}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1)); name=pmc_g$i
  rm -rf /tmp/rp_$name
  ( cd $REPO && timeout 300 rocprofv3 --kernel-trace --pmc $group -d /tmp/rp_$name -o $name --output-format csv -- python scratch/iter_probe.py ) > $OUT/$name.log 2>&1
  echo "== $name ($group) rc=$?"
  find /tmp/rp_$name -name "*counter_collection.csv" | while read f; do cp "$f" $OUT/$(basename "$f"); done
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GROUPS
python - <<PY > $OUT/per_iter.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        if "search_kernel" not in r["Kernel_Name"] or "icp_" not in r["Kernel_Name"]: continue
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(per)
    print("#", f.split("/")[-1])
    names = sorted(per[ids[0]]) if ids else []
    print("%-4s %9s " % ("it", "us") + " ".join("%16s" % n[-16:] for n in names))
    for k, d in enumerate(ids):
        print("%-4d %9.1f " % (k % 6, dur[d]) + " ".join("%16.5g" % per[d][n] for n in names))
PY
cat $OUT/per_iter.txt
