#!/bin/bash
# round 6, GPU call 1: producer waves -- parity first, then ablation stamps, then the A/B
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_persist_producers.py -x -q > $O/call1_pytest.txt 2>&1 || { tail -30 $O/call1_pytest.txt; exit 1; }
tail -3 $O/call1_pytest.txt
for b in 0 notail 2 16; do
  for B in 1000 100; do timeout -k 10 120 tools/persist_ablate_$b $B >> $O/persist_ablate.txt 2>&1 || exit 1; done
done
echo ABLATE_DONE
timeout -k 10 1500 tools/ab_persist_pw.sh > $O/ab_persist_pw.txt 2>&1 || { tail -20 $O/ab_persist_pw.txt; exit 1; }
echo CALL1_DONE
