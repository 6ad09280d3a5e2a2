#!/bin/bash
# Same-box A/B (developer tool, round 6): the five-waves-side-by-side shape (256 < N <= 320, DL and Langevin) with equal K
# halves (CCVM_AMD_PERSIST_XS=0) against the unequal split that balances the SIMDs' issue (default).
#   usage: bash tools/ab_persist_xs.sh > gpurun_out/r06/ab_persist_xs.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for k in dl langevin; do for n in 257 272 288 300 304 320; do for b in 128 512 1000 2000 4000; do CASES="$CASES $k:$n:$b"; done; done; done
CASES="$CASES pl:300:1000"
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
export CCVM_AMD_PERSIST_WIDE=1
for rep in 1 2; do
export CCVM_AMD_PERSIST_XS=0; run "equal"
export CCVM_AMD_PERSIST_XS=1; run "unequal"
done
echo AB_DONE
