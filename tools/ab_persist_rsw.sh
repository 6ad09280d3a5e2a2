#!/bin/bash
# Same-box sweep (developer tool, round 6): 128 < N <= 192, batches beyond one row set per CU -- the default plan against
# whole chains (KH=1), six-wave workgroups (KH=2 RSW=1) and TWO six-wave row sets per workgroup (KH=2 RSW=2).
#   usage: bash tools/ab_persist_rsw.sh > gpurun_out/r06/ab_persist_rsw.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for k in dl langevin mf; do for n in 130 144 160 176 192; do for b in 300 512 600 768 1000 1500 2000 3000 4000; do CASES="$CASES $k:$n:$b"; done; done; done
for k in langevin mf; do for n in 144 160 192; do for b in 768 1000 2000 4000; do CASES="$CASES $k:$n:$b:adam"; done; done; done
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
unset CCVM_AMD_PERSIST_KH CCVM_AMD_PERSIST_RSW; run "default"
export CCVM_AMD_PERSIST_KH=1; run "KH=1"
export CCVM_AMD_PERSIST_KH=2 CCVM_AMD_PERSIST_RSW=1; run "KH=2,RSW=1"
export CCVM_AMD_PERSIST_KH=2 CCVM_AMD_PERSIST_RSW=2; run "KH=2,RSW=2"
echo AB_DONE
