#!/bin/bash
# round 6, call 25: two row sets per twelve-wave workgroup as the default where it pays -- the whole GPU suite, then the
# cells of 128 < N <= 192 again under the default policy
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_rsw.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_rsw.txt; exit 1; }
tail -2 $O/gpu_suite_rsw.txt
CASES=""
for k in dl langevin mf; do for n in 130 144 160 176 192; do for b in 256 512 600 768 1000 1500 2000 3000 4000; do CASES="$CASES $k:$n:$b"; done; done; done
for k in langevin mf; do for n in 144 160 192; do for b in 512 1000 2000 4000; do CASES="$CASES $k:$n:$b:adam"; done; done; done
timeout -k 10 200 python3 tools/time_small.py $CASES > $O/persist_129_192_default.txt 2>&1 || { echo TIME_FAILED; tail -5 $O/persist_129_192_default.txt; exit 1; }
tail -2 $O/persist_129_192_default.txt
echo CALL25_DONE
