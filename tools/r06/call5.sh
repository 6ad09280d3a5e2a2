#!/bin/bash
# round 6, GPU call 5: direct launch vs ABI at N = 20; exchange kernels with step-scaled bounds and one launch per round
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 120 tools/persist_vs_abi 1000 > $O/persist_vs_abi.txt 2>&1 || { tail $O/persist_vs_abi.txt; exit 1; }
cat $O/persist_vs_abi.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_slab.py tests/test_gpu_ptile.py -x -q > $O/call5_pytest.txt 2>&1 || { tail -40 $O/call5_pytest.txt; exit 1; }
tail -3 $O/call5_pytest.txt
for w in mf_n500_b1000 langevin_n500_b1000 dl_n1000_b32; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail -20 $O/bench_$w.err; exit 1; }
done
echo CALL5_DONE
