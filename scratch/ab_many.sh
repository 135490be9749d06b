#!/bin/bash
# scratch/ab_many.sh <tag> "<variants>" : default + variants over config 3 (sheet, families), config 2 and the k-NN probe
TAG=$1; VARS=$2; OUT=gpurun_out/$TAG; mkdir -p $OUT
for a in "--config 3" "--config 2" "--cloud cube" "--cloud layers" "--cloud clusters"; do
  echo "== $a" | tee -a $OUT/ab.log
  for v in default $VARS; do
    L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
    PCLHIP_LIB=$L timeout 300 python bench.py $a --no-cpu-baseline --no-host-align --no-families > $OUT/$v.json 2> $OUT/$v.err
    python scratch/ab_line.py $v $OUT/$v.json | tee -a $OUT/ab.log
  done
done
for v in default $VARS; do
  L=pcl_amd/libpclhip.so; [ $v != default ] && L=pcl_amd/variants/libpclhip_$v.so
  echo "== knn_probe $v" | tee -a $OUT/ab.log
  PCLHIP_LIB=$L python scratch/knn_probe.py 10000000 2>&1 | grep "kernel ms" | tee -a $OUT/ab.log
done
