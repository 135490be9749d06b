#!/bin/bash
# timing probe of kd_block_kernel's parts (GPU box): variants that skip the four-way / the binary levels (results invalid)
mkdir -p gpurun_out/kdb
for v in 0 1 2 3; do
  L=pcl_amd/libpclhip.so; [ $v != 0 ] && L=pcl_amd/variants/libpclhip_kdb$v.so
  N=10000000 bash scratch/prof_build.sh kdb/v$v $L | grep -E "kd_block|kd_finish|sum of" > gpurun_out/kdb/v$v.txt
  echo "variant $v:"; cat gpurun_out/kdb/v$v.txt
done
