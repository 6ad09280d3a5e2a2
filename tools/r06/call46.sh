#!/bin/bash
# round 6, call 46: the unequal K split of the six-wave workgroups (three side by side) -- parity (the K-split tests force
# it at small batches), then the A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider > $O/xs3_tests.txt 2>&1; echo "ksplit tests exit $?"; tail -4 $O/xs3_tests.txt
timeout -k 10 500 bash tools/ab_persist_xs3.sh > $O/ab_persist_xs3.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_xs3.txt; exit 1; }
tail -2 $O/ab_persist_xs3.txt
echo CALL46_DONE
