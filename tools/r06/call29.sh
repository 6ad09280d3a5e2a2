#!/bin/bash
# round 6, call 29: the N = 70 bench line again, with its new kernel's counter summary in place (the line's model block)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
w=dl_n70_b1000
timeout -k 10 200 python bench.py --workload $w > gpurun_out/r06_bench_$w.json 2> $O/bench_$w.err || { echo BENCH_FAILED; tail -5 $O/bench_$w.err; exit 1; }
tail -c 600 gpurun_out/r06_bench_$w.json
echo CALL29_DONE
