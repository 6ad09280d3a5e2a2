#!/bin/bash
# round 6, call 21: the K split at small batches (N > 64), both column-group variants
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 bash tools/ab_persist_kh_small.sh > $O/ab_persist_kh_small.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_kh_small.txt; exit 1; }
tail -2 $O/ab_persist_kh_small.txt
echo CALL21_DONE
