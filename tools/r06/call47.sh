#!/bin/bash
# round 6, call 47: the six-wave workgroups' short part swept
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 1000 bash tools/ab_persist_xs3_delta.sh > $O/ab_persist_xs3_delta.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_xs3_delta.txt; exit 1; }
tail -2 $O/ab_persist_xs3_delta.txt
echo CALL47_DONE
