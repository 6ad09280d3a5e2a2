#!/bin/bash
# round 6, call 39: MF (N <= 272) and Langevin + Adam (N <= 288) on the five-wave shape -- the whole GPU suite, the audit's MF
# cells of N = 257 again
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_wide2.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_wide2.txt; exit 1; }
tail -1 $O/gpu_suite_wide2.txt
rm -f $O/policy_regret_wide2.jsonl
timeout -k 10 300 python3 tools/policy_regret.py --out $O/policy_regret_wide2.jsonl --kinds mf --ns 257 --bs 32,128,256,512,768,1000,1500,2000 > $O/policy_regret_wide2.log 2>&1 || { echo REGRET_STOPPED; tail -3 $O/policy_regret_wide2.log; }
tail -3 $O/policy_regret_wide2.log
echo CALL39_DONE
