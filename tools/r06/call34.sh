#!/bin/bash
# round 6, call 34: the size sweep of the final code again (N = 257 ... 320 moved) and the soak with the five-wave shape
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 420 bash tools/size_sweep.sh > $O/size_sweep.log 2>&1 || { echo SWEEP_FAILED; tail -5 $O/size_sweep.log; exit 1; }
cp gpurun_out/size_sweep_auto.txt $O/size_sweep_auto.txt; cp gpurun_out/size_sweep_tile.txt $O/size_sweep_tile.txt
timeout -k 10 500 python tools/soak_shapes.py > $O/soak_shapes.txt 2>&1 || { echo SOAK_FAILED; tail -8 $O/soak_shapes.txt; exit 1; }
tail -5 $O/soak_shapes.txt | cut -c1-160
echo CALL34_DONE
