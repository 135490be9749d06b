#!/bin/bash
# A/B builds: scripts/build_variant.sh <name> "<extra hipcc flags>" [source.hip] -> pcl_amd/variants/libpclhip_<name>.so
# (only that source -- search.hip by default -- is recompiled with the flags; the other objects come from the default
# build).  Use with PCLHIP_LIB=pcl_amd/variants/libpclhip_<name>.so (scripts/ab_variants.sh runs the bench over them).
set -e
name=$1; flags=$2; src=${3:-search.hip}
obj=${src%.*}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}; ARCH=${ARCH:-gfx950}   # the same overrides the main Makefile honours
cd "$(dirname "$0")/../pcl_amd/csrc"
make -j8 HIPCC="$HIPCC" ARCH="$ARCH" >/dev/null
mkdir -p ../variants
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=$ARCH -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result \
  -I../../include -I. $flags -c $src -o ../variants/${obj}_$name.o
objs=$(ls *.o | grep -v "^$obj.o$")
$HIPCC --offload-arch=$ARCH -shared -fPIC -o ../variants/libpclhip_$name.so ../variants/${obj}_$name.o $objs -ldl
echo built pcl_amd/variants/libpclhip_$name.so
