#!/bin/bash
# round 6, call 22: does forcing the row-owner workgroups apart (LDS padding) pay?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 bash tools/ab_persist_spread.sh > $O/ab_persist_spread.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_spread.txt; exit 1; }
tail -2 $O/ab_persist_spread.txt
echo CALL22_DONE
