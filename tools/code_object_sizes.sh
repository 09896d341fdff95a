#!/bin/bash
# Device code object of every translation unit: bytes (uncompressed ELF) and kernels (symbols ending in .kd) — what the runtime loads
# on a unit's first launch.  Usage: bash tools/code_object_sizes.sh [unit ...]   (runs anywhere: hipcc cross-compiles)
cd "$(dirname "$0")/../colorid_amd/csrc"
B=/opt/rocm/lib/llvm/bin
units=${@:-$(ls *.hip | sed 's/\.hip$//')}
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c $u.hip -o /tmp/co_$u.co 2>/dev/null || { echo "$u: compile failed"; continue; }
  $B/clang-offload-bundler --unbundle --type=o --input=/tmp/co_$u.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/co_$u.elf
  k=$($B/llvm-readelf -s /tmp/co_$u.elf 2>/dev/null | grep -c '\.kd$')
  echo "$u $(stat -c %s /tmp/co_$u.elf) bytes $k kernels"
  rm -f /tmp/co_$u.co /tmp/co_$u.elf
done
