#!/bin/bash
# round 6, call 37: the N = 300 bench lines again, with their kernels' counter summaries in place (the lines' model blocks)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
for w in dl_n300_b1000 langevin_n300_b1000; do
  timeout -k 10 200 python bench.py --workload $w > gpurun_out/r06_bench_$w.json 2> $O/bench_$w.err || { echo BENCH_FAILED $w; tail -5 $O/bench_$w.err; exit 1; }
done
echo CALL37_DONE
