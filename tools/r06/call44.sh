#!/bin/bash
# round 6, call 44: parity of the unequal K split (the five-wave tests), then its short part's length swept
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_persist_ksplit.py -q -m gpu -p no:cacheprovider -k "five_waves or other_variants" > $O/xs_tests.txt 2>&1; echo "xs tests exit $?"; tail -3 $O/xs_tests.txt
timeout -k 10 900 bash tools/ab_persist_xs_delta.sh > $O/ab_persist_xs_delta.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_xs_delta.txt; exit 1; }
tail -2 $O/ab_persist_xs_delta.txt
echo CALL44_DONE
