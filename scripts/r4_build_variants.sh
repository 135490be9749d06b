#!/bin/bash
# Builds, in the CPU container, every variant scripts/r4_first_call.sh measures (they travel to the GPU box with the snapshot):
# pcl_amd/variants/libpclhip_<name>.so.  ~75 s per variant, four at a time.
set -e
cd "$(dirname "$0")/.."
make -C pcl_amd/csrc -j8 > /dev/null
build() { bash scripts/build_variant.sh "$1" "$2" > /dev/null 2>&1 && echo "built $1" || echo "FAILED $1"; }
build run3 "-DPCLHIP_COLD_RUN=3" &
build run6 "-DPCLHIP_COLD_RUN=6" &
build greedy "-DPCLHIP_SO_GREEDY_SEED=1" &
build greedy_r2 "-DPCLHIP_SO_GREEDY_SEED=1 -DPCLHIP_COLD_RUN=2" &
wait
build greedy_r1 "-DPCLHIP_SO_GREEDY_SEED=1 -DPCLHIP_COLD_RUN=1" &
build grec "-DPCLHIP_GROUP_LISTS=1" &
build grec103 "-DPCLHIP_GROUP_LISTS=1 -DPCLHIP_GREC_GROW=1.03f" &
build greedy_r1_grec "-DPCLHIP_SO_GREEDY_SEED=1 -DPCLHIP_COLD_RUN=1 -DPCLHIP_GROUP_LISTS=1" &
wait
ls -la pcl_amd/variants/*.so
