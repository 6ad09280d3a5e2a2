#!/bin/bash
# Same-box sweep (developer tool, round 6): 64 < N <= 96 -- the default (two 64-column waves side by side: K split /
# producers by the policy) against THREE 32-column waves of eight rows, whole chains, two / four rows in use, with / without
# noise producer waves.     usage: bash tools/ab_persist_cw32.sh > gpurun_out/r06/ab_persist_cw32.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for k in dl langevin mf; do for n in 65 70 80 96; do for b in 32 100 256 500 1000 1500 2000 3000 4000 8000; do CASES="$CASES $k:$n:$b"; done; done; done
for k in langevin mf; do for n in 70 96; do for b in 100 1000 4000; do CASES="$CASES $k:$n:$b:adam"; done; done; done
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
unset CCVM_AMD_PERSIST_CW CCVM_AMD_PERSIST_RU CCVM_AMD_PERSIST_PW; run "default"
export CCVM_AMD_PERSIST_CW=32
for ru in 2 4; do for pw in 0 1; do export CCVM_AMD_PERSIST_RU=$ru CCVM_AMD_PERSIST_PW=$pw; run "narrow,RU=$ru,PW=$pw"; done; done
echo AB_DONE
