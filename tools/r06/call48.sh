#!/bin/bash
# round 6, call 48: the unequal K splits as defaults (six-wave and ten-wave workgroups) -- the whole GPU suite, the soak of
# the shapes, the size sweep's row-owner part, the dl_n160 / n300 bench lines
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_xs3.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_xs3.txt; exit 1; }
tail -1 $O/gpu_suite_xs3.txt
timeout -k 10 500 python tools/soak_shapes.py > $O/soak_shapes.txt 2>&1 || { echo SOAK_FAILED; tail -8 $O/soak_shapes.txt; exit 1; }
tail -1 $O/soak_shapes.txt
CASES=""
for k in dl langevin mf; do for n in 130 144 160 176 192; do for b in 256 512 768 1000 2000; do CASES="$CASES $k:$n:$b"; done; done; done
timeout -k 10 200 python3 tools/time_small.py $CASES > $O/persist_129_192_default.txt 2>&1 || { echo TIME_FAILED; exit 1; }
echo CALL48_DONE
