#!/bin/bash
# Generic PMC profiling of a command on the GPU box.  Usage: scripts/profile_cmd.sh <tag> -- <command...>
# One rocprofv3 run per counter group (with --kernel-trace only).
set -u
TAG=$1; shift; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1)); name=pmc_g$i
  rm -rf /tmp/rp_$name
  ( cd $REPO && timeout 600 rocprofv3 --kernel-trace --pmc $group -d /tmp/rp_$name -o $name --output-format csv -- "$@" ) > $OUT/$name.log 2>&1
  echo "== $name ($group) rc=$?"
  find /tmp/rp_$name -name "*counter_collection.csv" | while read f; do cp "$f" $OUT/$(basename "$f"); done
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A12 "icp_iterate" $OUT/summary.txt | grep -v "^--"
