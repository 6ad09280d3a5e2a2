#!/bin/bash
# round 6, call 18: size sweep of the final code (small sizes included) and a regret audit on this round's box
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 420 bash tools/size_sweep.sh > $O/size_sweep.log 2>&1 || { echo SWEEP_FAILED; tail -5 $O/size_sweep.log; exit 1; }
cp gpurun_out/size_sweep_auto.txt $O/size_sweep_auto.txt; cp gpurun_out/size_sweep_tile.txt $O/size_sweep_tile.txt
timeout -k 10 700 python3 tools/policy_regret.py --out $O/policy_regret.jsonl --ns 100,256,300,500,640,768,1000,1500,2000 --bs 32,256,512,768,1000,2000 > $O/policy_regret.log 2>&1 || { echo REGRET_STOPPED; tail -3 $O/policy_regret.log; }
echo CALL18_DONE
