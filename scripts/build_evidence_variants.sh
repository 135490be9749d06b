#!/bin/bash
# The instrumented builds scripts/final_evidence.sh runs (pcl_amd/variants/, not tracked): build them before the gpurun call.
set -e
cd "$(dirname "$0")/.."
bash scripts/build_variant.sh stats "-DPCLHIP_SO_STATS"
bash scripts/build_variant.sh prof "-DPCLHIP_SO_PROFILE"
bash scripts/build_variant.sh why "-DPCLHIP_SO_REASONS"
bash scripts/build_variant.sh nrmprof "-DPCLHIP_NRM_PROFILE"
bash scripts/build_variant.sh icpprof "-DPCLHIP_ICP_PROFILE"
bash scripts/build_variant.sh lanes "-DPCLHIP_STATS_LANES"
bash scripts/build_variant.sh rec1 "-DPCLHIP_REC_CAP=1"
bash scripts/build_variant.sh kdbticks "-DPCLHIP_KDB_TICKS" index_build.hip
