#!/bin/bash
# round 6, call 50: the unequal K split for MF (N <= 272) and Langevin + Adam (N <= 288) too -- parity, then equal halves
# against it
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "five_waves or other_variants" > $O/xs2_tests.txt 2>&1; echo "tests exit $?"; tail -3 $O/xs2_tests.txt
CASES=""
for n in 257 272; do for b in 512 1000 2000; do CASES="$CASES mf:$n:$b"; done; done
for n in 257 272 288; do for b in 512 1000 2000; do CASES="$CASES langevin:$n:$b:adam"; done; done
run() { python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$1 /" || exit 1; }
export CCVM_AMD_PERSIST_WIDE=1
: > $O/ab_persist_xs2.txt
for rep in 1 2; do
export CCVM_AMD_PERSIST_XS=0; run "equal" >> $O/ab_persist_xs2.txt
export CCVM_AMD_PERSIST_XS=1; run "unequal" >> $O/ab_persist_xs2.txt
done
tail -2 $O/ab_persist_xs2.txt
echo CALL50_DONE
