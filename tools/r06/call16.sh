#!/bin/bash
# round 6, GPU call 16: host side of a solver call (schedule cache, one status read), the whole GPU suite, TTS again
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python tools/solve_cost.py dl > $O/solve_cost_dl_after.txt 2>&1 || { tail -20 $O/solve_cost_dl_after.txt; exit 1; }
grep "__call__ wall" $O/solve_cost_dl_after.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/call16_pytest.txt 2>&1; rc=$?
tail -6 $O/call16_pytest.txt | cut -c1-250
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/tts_report.py --out $O/tts.json > $O/tts.txt 2> $O/tts.err || { tail -20 $O/tts.err; exit 1; }
cat $O/tts.txt
echo CALL16_DONE
