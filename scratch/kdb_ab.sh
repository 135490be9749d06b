#!/bin/bash
# A/B of kd_block_kernel variants (GPU box): per-dispatch time of the last build + the build's own figure
mkdir -p gpurun_out/kdb
for v in default kdbunpacked kdbw8; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  N=10000000 bash scratch/prof_build.sh kdb/$v $L | grep -E "^build|kd_block|sum of" | tail -3 > gpurun_out/kdb/$v.txt
  echo "variant $v:"; cat gpurun_out/kdb/$v.txt
done
PCLHIP_LIB=pcl_amd/variants/libpclhip_kdbticks.so python scratch/kdb_ticks.py
