#!/bin/bash
# round 6, call 45: the unequal K split as the default of the five-wave shape -- the whole GPU suite, the audit's DL / Langevin
# cells of N = 257 / 300 / 320 again, the two N = 300 bench lines
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_xs.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_xs.txt; exit 1; }
tail -1 $O/gpu_suite_xs.txt
rm -f $O/policy_regret_xs.jsonl
timeout -k 10 400 python3 tools/policy_regret.py --out $O/policy_regret_xs.jsonl --kinds dl,langevin --ns 257,300,320 --bs 32,128,256,512,768,1000,1500,2000 > $O/policy_regret_xs.log 2>&1 || { echo REGRET_STOPPED; tail -3 $O/policy_regret_xs.log; }
tail -2 $O/policy_regret_xs.log
for w in dl_n300_b1000 langevin_n300_b1000; do
  timeout -k 10 200 python bench.py --workload $w > gpurun_out/r06_bench_$w.json 2> $O/bench_$w.err || { echo BENCH_FAILED $w; tail -5 $O/bench_$w.err; exit 1; }
done
echo CALL45_DONE
