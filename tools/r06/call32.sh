#!/bin/bash
# round 6, call 32: 256 < N <= 320 on the row-owner kernel (five waves side by side, fragments partly in LDS) -- parity, then
# the sweep against what ran before
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "five_waves or other_variants" > $O/wide_tests.txt 2>&1; echo "wide tests exit $?"; tail -12 $O/wide_tests.txt
timeout -k 10 600 bash tools/ab_persist_wide.sh > $O/ab_persist_wide.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_wide.txt; exit 1; }
tail -2 $O/ab_persist_wide.txt
echo CALL32_DONE
