#!/bin/bash
# scripts/device_code_hash.sh <object.o> [...] : sha256 of the gfx950 .text of every object (the device instructions only;
# metadata, symbol order of host stubs and build paths do not enter).  Two builds whose kernels disassemble identically
# print the same hash -- used to prove that a clean-up of #if arms left the shipped kernels untouched.
B=/opt/rocm/lib/llvm/bin
for o in "$@"; do
  t=$(mktemp -d)
  $B/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$o" 2>/dev/null
  $B/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.o 2>/dev/null
  $B/llvm-objcopy --dump-section .text=$t/text.bin $t/dev.o 2>/dev/null
  echo "$(sha256sum < $t/text.bin | cut -c1-16)  $(stat -c %s $t/text.bin) bytes  $o"
  rm -rf $t
done
