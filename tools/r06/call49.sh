#!/bin/bash
# round 6, call 49: the size sweep of the final code once more (the unequal K splits moved N = 257 ... 320)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 420 bash tools/size_sweep.sh > $O/size_sweep.log 2>&1 || { echo SWEEP_FAILED; tail -5 $O/size_sweep.log; exit 1; }
cp gpurun_out/size_sweep_auto.txt $O/size_sweep_auto.txt; cp gpurun_out/size_sweep_tile.txt $O/size_sweep_tile.txt
echo CALL49_DONE
