#!/bin/bash
# round 6, call 30: soak of the row-owner kernel's new shapes against the ones they replaced
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 500 python tools/soak_shapes.py > $O/soak_shapes.txt 2>&1 || { echo SOAK_FAILED; tail -8 $O/soak_shapes.txt; exit 1; }
cat $O/soak_shapes.txt | cut -c1-200
echo CALL30_DONE
