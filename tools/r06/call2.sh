#!/bin/bash
# round 6, GPU call 2: producer waves with the role swap -- parity, ablation at three batch sizes, the A/B
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_persist_producers.py -x -q > $O/call2_pytest.txt 2>&1 || { tail -30 $O/call2_pytest.txt; exit 1; }
tail -3 $O/call2_pytest.txt
for B in 1000 2000 500; do timeout -k 10 120 tools/persist_ablate_0 $B >> $O/persist_ablate_swap.txt 2>&1 || exit 1; done
echo ABLATE_DONE
timeout -k 10 1500 tools/ab_persist_pw.sh > $O/ab_persist_pw.txt 2>&1 || { tail -20 $O/ab_persist_pw.txt; exit 1; }
echo CALL2_DONE
