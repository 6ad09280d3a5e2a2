#!/bin/bash
# Where the column-cluster kernel's step goes at config 3's shape (N = 500, B = 1000), Langevin and MF, on the code as it is
# (developer tool; VERDICT r4 item 1's alternative done-criterion): the full kernel, without its MFMAs, without its noise,
# without its exchange, and the s_memtime stamps of a phase's segments.   usage: tools/cluster_floor.sh > out.txt
cd "$(dirname "$0")/.."
for mode in "" "-DCCVM_ABL_MF"; do
  for b in 0 1 2 8 64; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $mode -DCCVM_CLUSTER_ABL=$b tools/cluster_ablate.hip -o /tmp/cluster_ablate_$b || exit 1
  done
  for rep in 1 2; do for b in 0 1 2 8; do /tmp/cluster_ablate_$b 500 1000; done; done
  /tmp/cluster_ablate_64 500 1000
done
