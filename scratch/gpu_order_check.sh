#!/bin/bash
# the kd order of the current library against the saved base build, bit for bit, on the GPU; then per-dispatch times
OUT=gpurun_out/$1; mkdir -p $OUT
SIZES="10 17 100 4096 4097 5000 16385 20000 70000 270000 1100000 3000000 10000000"
PCLHIP_LIB=pcl_amd/libpclhip.so timeout 200 python scratch/order_dump.py $OUT/new.npz $SIZES 2>&1 | tail -2
PCLHIP_LIB=pcl_amd/variants/libpclhip_base.so timeout 200 python scratch/order_dump.py $OUT/base.npz $SIZES 2>&1 | tail -2
python scratch/cmp_orders.py $OUT/base.npz $OUT/new.npz | tee $OUT/cmp.txt
rm -f $OUT/new.npz $OUT/base.npz
bash scratch/prof_build.sh $1
