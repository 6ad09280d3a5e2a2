#!/bin/bash
# Same-box A/B (developer tool, round 6): three waves side by side x two K halves in ONE six-wave workgroup per CU (128 < N <= 192,
# batches of up to a row set per CU above the two-rows rule's range) -- equal halves (CCVM_AMD_PERSIST_XS=0) against the
# unequal split whose long parts run on the waves that have a SIMD to themselves (default).
#   usage: bash tools/ab_persist_xs3.sh > gpurun_out/r06/ab_persist_xs3.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for n in 130 144 160 176 192; do for b in 300 400 512; do CASES="$CASES dl:$n:$b"; done; for b in 600 768 1000 1024; do CASES="$CASES langevin:$n:$b mf:$n:$b"; done; done
for n in 144 176; do for b in 768 1000; do CASES="$CASES langevin:$n:$b:adam mf:$n:$b:adam"; done; done
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
for rep in 1 2; do
export CCVM_AMD_PERSIST_XS=0; run "equal"
export CCVM_AMD_PERSIST_XS=1; run "unequal"
done
echo AB_DONE
