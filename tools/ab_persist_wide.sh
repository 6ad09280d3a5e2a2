#!/bin/bash
# Same-box sweep (developer tool, round 6): 256 < N <= 320, DL and Langevin -- the row-owner kernel's five waves side by side
# (default) against what ran before (CCVM_AMD_PERSIST_WIDE=0: column-cluster / slab / tile kernels by the launch policy).
#   usage: bash tools/ab_persist_wide.sh > gpurun_out/r06/ab_persist_wide.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for k in dl langevin; do for n in 257 272 288 300 304 320; do for b in 1 32 128 256 512 768 1000 1500 2000 4000; do CASES="$CASES $k:$n:$b"; done; done; done
CASES="$CASES pl:300:1000 pl:320:512"
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
export CCVM_AMD_PERSIST_WIDE=0; run "before"
unset CCVM_AMD_PERSIST_WIDE; run "five-waves"
echo AB_DONE
