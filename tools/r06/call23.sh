#!/bin/bash
# round 6, call 23: three waves side by side + the small-batch two-row rule as the default -- the whole GPU suite, then the
# affected cells again (default policy, this build) and the size sweep's row-owner part
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_ncg3.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_ncg3.txt; exit 1; }
tail -2 $O/gpu_suite_ncg3.txt
CASES=""
for k in dl langevin mf; do for n in 130 144 160 176 192 200 224 240 256; do for b in 64 256 512 768 1000 2000; do CASES="$CASES $k:$n:$b"; done; done; done
for k in langevin mf; do for n in 144 192 224; do for b in 256 512 1000; do CASES="$CASES $k:$n:$b:adam"; done; done; done
timeout -k 10 300 python3 tools/time_small.py $CASES > $O/persist_above128_default.txt 2>&1 || { echo TIME_FAILED; tail -5 $O/persist_above128_default.txt; exit 1; }
tail -2 $O/persist_above128_default.txt
echo CALL23_DONE
