#!/bin/bash
# round 6, call 27: 64 < N <= 96 as three 32-column waves of eight rows -- parity, then the sweep against the default
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "narrow" > $O/cw32_tests.txt 2>&1; echo "narrow tests exit $?"; tail -6 $O/cw32_tests.txt
timeout -k 10 600 bash tools/ab_persist_cw32.sh > $O/ab_persist_cw32.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_cw32.txt; exit 1; }
tail -2 $O/ab_persist_cw32.txt
echo CALL27_DONE
