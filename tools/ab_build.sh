#!/bin/bash
# usage: tools/ab_build.sh "<extra hipcc flags>" -- on the GPU box: build a variant of the library with extra compiler
# flags into /tmp/libccvm_variant.so (the in-tree library stays what was shipped), for same-box A/B runs through
# CCVM_AMD_LIB (ccvm_amd/_lib.py).
set -e
rm -rf /tmp/ccvm_variant && mkdir -p /tmp/ccvm_variant
cd ccvm_amd/csrc
for u in *.hip; do
  f=""; case $u in ccvm_persist_*) f="-mllvm -amdgpu-mfma-vgpr-form";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f $1 -c $u -o /tmp/ccvm_variant/${u%.hip}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ccvm_variant/*.o -o /tmp/libccvm_variant.so
echo built /tmp/libccvm_variant.so
