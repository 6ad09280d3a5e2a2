#!/bin/bash
# round 6, call 38: the five-wave shape for MF (N <= 272) and Langevin + Adam (N <= 288) -- parity, then what a round costs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "five_waves or other_variants" > $O/wide2_tests.txt 2>&1; echo "wide tests exit $?"; tail -8 $O/wide2_tests.txt
CASES=""
for n in 257 272; do for b in 1 32 128 256 512 1000 1024 1500 2000 4000; do CASES="$CASES mf:$n:$b"; done; done
for n in 257 272 288; do for b in 1 32 128 256 512 1000 1024 1500 2000 4000; do CASES="$CASES langevin:$n:$b:adam"; done; done
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
export CCVM_AMD_PERSIST_WIDE=0; run "before" > $O/ab_persist_wide2.txt
export CCVM_AMD_PERSIST_WIDE=1; run "five-waves" >> $O/ab_persist_wide2.txt
tail -2 $O/ab_persist_wide2.txt
echo CALL38_DONE
