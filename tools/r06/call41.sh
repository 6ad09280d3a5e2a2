#!/bin/bash
# round 6, call 41: where the GPU suite's time goes
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=40 > $O/gpu_suite_durations.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_durations.txt; exit 1; }
grep -A45 "slowest 40 durations" $O/gpu_suite_durations.txt | cut -c1-160
tail -1 $O/gpu_suite_durations.txt
echo CALL41_DONE
