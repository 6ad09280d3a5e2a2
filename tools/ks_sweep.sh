#!/bin/bash
# Tile shape of the per-step kernel: 32 x 128 tiles (KS=1) against 32 x 64 split-K tiles (KS=2) per size.
# usage (GPU box): tools/ks_sweep.sh > gpurun_out/ks_sweep.txt
cases="$*"
[ -z "$cases" ] && cases="dl:896:1000 dl:1000:1000 dl:1200:1000 dl:1300:1000 dl:1500:1000 dl:1700:1000 dl:2000:1000 dl:2500:1000 dl:3000:1000 dl:1000:500 dl:1500:500 dl:2000:500 dl:1000:2000 dl:1500:2000 langevin:896:1000 langevin:1000:1000 langevin:1200:1000 langevin:1500:1000 langevin:1700:1000 langevin:2000:1000 langevin:3000:1000 langevin:1500:500 langevin:2000:512 mf:1000:1000 mf:1500:1000"
for ks in 1 2; do
  echo "== CCVM_AMD_KS=$ks"
  CCVM_AMD_KS=$ks CCVM_AMD_KERNEL=tile python3 tools/time_small.py $cases || exit 1
done
