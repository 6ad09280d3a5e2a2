#!/bin/bash
# the cluster kernel's fetch waves making the next phase's normals (-DCCVM_CL_FETCH_NOISE=1) against the
# MFMA waves making their own (=0), config 3's shape: per-step time and the s_memtime stamps of a phase's segments
# (developer tool).   usage: tools/cluster_fetch_noise.sh > gpurun_out/r06/cluster_fetch_noise.txt
cd "$(dirname "$0")/.."
for mode in "" "-DCCVM_ABL_MF"; do
  for fn in 0 1; do
    for b in 0 64; do
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $mode -DCCVM_CLUSTER_ABL=$b -DCCVM_CL_FETCH_NOISE=$fn tools/cluster_ablate.hip -o /tmp/cluster_fn${fn}_$b || exit 1
    done
  done
  for rep in 1 2; do for fn in 0 1; do echo "CCVM_CL_FETCH_NOISE=$fn"; /tmp/cluster_fn${fn}_0 500 1000; done; done
  for fn in 0 1; do echo "CCVM_CL_FETCH_NOISE=$fn, stamps"; /tmp/cluster_fn${fn}_64 500 1000; done
done
echo "--- a dropped workgroup: the launch's duration with a 20 ms bound (2 000 000 ticks)"
CL_DROP=1 CL_SPIN=2000000 /tmp/cluster_fn0_0 500 1000
