#!/bin/bash
# round 6, call 52: DL / Langevin at three side by side take the six-wave unequal split from one row on -- the GPU suite, the
# soak of the shapes, the small-batch cells again
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_final.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_final.txt; exit 1; }
tail -1 $O/gpu_suite_final.txt
timeout -k 10 500 python tools/soak_shapes.py > $O/soak_shapes.txt 2>&1 || { echo SOAK_FAILED; tail -8 $O/soak_shapes.txt; exit 1; }
tail -1 $O/soak_shapes.txt
CASES=""
for k in dl langevin mf; do for n in 130 144 160 176 192; do for b in 1 64 256 512 768 1000 2000; do CASES="$CASES $k:$n:$b"; done; done; done
timeout -k 10 200 python3 tools/time_small.py $CASES > $O/persist_129_192_default.txt 2>&1 || { echo TIME_FAILED; exit 1; }
echo CALL52_DONE
