#!/bin/bash
# round 6, GPU call 10: fetch-wave noise stamps (the negative result), what a dropped workgroup costs, sc1 stress test
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 tools/cluster_fetch_noise.sh > $O/cluster_fetch_noise.txt 2>&1 || { tail -20 $O/cluster_fetch_noise.txt; exit 1; }
grep -E "us/step|noise \+ update|wait at B_0|end to end|FETCH_NOISE" $O/cluster_fetch_noise.txt | cut -c1-170
timeout -k 10 600 python tools/fault_cost.py > $O/fault_cost.txt 2>&1 || { tail -20 $O/fault_cost.txt; exit 1; }
cat $O/fault_cost.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ptile.py -q -k "sc1_lds_dma or library_made" > $O/call10_pytest.txt 2>&1 || { tail -30 $O/call10_pytest.txt; exit 1; }
tail -2 $O/call10_pytest.txt
echo CALL10_DONE
