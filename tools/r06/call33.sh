#!/bin/bash
# round 6, call 33: the row-owner kernel's five waves side by side as the default for DL / Langevin at 256 < N <= 320 -- the
# whole GPU suite; the audit's cells of N = 257 / 300 / 320 again (default and every forced plan); the two N = 300 bench lines
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_wide.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_wide.txt; exit 1; }
tail -2 $O/gpu_suite_wide.txt
rm -f $O/policy_regret_wide.jsonl
timeout -k 10 400 python3 tools/policy_regret.py --out $O/policy_regret_wide.jsonl --kinds dl,langevin,mf --ns 257,300,320 --bs 32,128,256,512,768,1000,1500,2000 > $O/policy_regret_wide.log 2>&1 || { echo REGRET_STOPPED; tail -3 $O/policy_regret_wide.log; }
tail -3 $O/policy_regret_wide.log
for w in dl_n300_b1000 langevin_n300_b1000; do
  timeout -k 10 200 python bench.py --workload $w > gpurun_out/r06_bench_$w.json 2> $O/bench_$w.err || { echo BENCH_FAILED $w; tail -5 $O/bench_$w.err; exit 1; }
done
echo CALL33_DONE
