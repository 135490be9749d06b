#!/bin/bash
# A/B of library variants over the bench (config 3) and the normals profile.  usage: scripts/r3_ab2.sh <tag> "<variants>"
tag=$1; out=gpurun_out/r3_$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in $2; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  echo "== $v" >> $out/ab.log
  PCLHIP_LIB=$L timeout 300 python scratch/ab.py 10 >> $out/ab.log 2>&1
done
if [ -f pcl_amd/variants/libpclhip_nrmprof.so ]; then
  echo "== normals profile (ticks per traverse call; groups counts two calls per group)" >> $out/ab.log
  PCLHIP_LIB=pcl_amd/variants/libpclhip_nrmprof.so timeout 300 python scratch/stats_probe.py 10000000 2>&1 | grep "normals" >> $out/ab.log
fi
cat $out/ab.log
