#!/bin/bash
# scripts/kernel_resources.sh <object.o> [...] : registers, spills, scratch and LDS of every gfx950 kernel in the objects
B=/opt/rocm/lib/llvm/bin
for o in "$@"; do
  t=$(mktemp -d)
  $B/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$o" 2>/dev/null
  $B/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.o 2>/dev/null
  $B/llvm-readelf --notes $t/dev.o | awk -v obj="$o" '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2}
    /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
    /\.wavefront_size:/ {printf "%-70s vgpr %3s sgpr %3s spill v%s s%s scratch %s lds %s\n", substr(name,1,70), v, s, vs, ss, p, g}'
  rm -rf $t
done
