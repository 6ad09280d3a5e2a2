#!/bin/bash
# round 6, call 43: the unequal K split of the five-wave shape -- parity (the wide tests run it by default), then the A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "five_waves or other_variants" > $O/xs_tests.txt 2>&1; echo "xs tests exit $?"; tail -6 $O/xs_tests.txt
CCVM_AMD_PERSIST_XS=0 timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "five_waves" > $O/xs0_tests.txt 2>&1; echo "equal-halves tests exit $?"; tail -3 $O/xs0_tests.txt
timeout -k 10 500 bash tools/ab_persist_xs.sh > $O/ab_persist_xs.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_xs.txt; exit 1; }
tail -2 $O/ab_persist_xs.txt
echo CALL43_DONE
