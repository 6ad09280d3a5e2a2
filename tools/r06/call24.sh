#!/bin/bash
# round 6, call 24: two six-wave row sets per workgroup (128 < N <= 192, K split) -- parity with the override on, then the sweep
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
CCVM_AMD_PERSIST_RSW=2 timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider > $O/rsw2_tests.txt 2>&1; echo "RSW=2 tests exit $?"; tail -4 $O/rsw2_tests.txt
timeout -k 10 700 bash tools/ab_persist_rsw.sh > $O/ab_persist_rsw.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_rsw.txt; exit 1; }
tail -2 $O/ab_persist_rsw.txt
echo CALL24_DONE
