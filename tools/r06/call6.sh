#!/bin/bash
# round 6, GPU call 6: ablate harness order / state, sc1 LDS-DMA stress, exchange kernels with time-based bounds
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
( timeout -k 10 60 tools/persist_ablate_0 1000 0; timeout -k 10 60 tools/persist_ablate_0 1000 1; timeout -k 10 60 tools/persist_ablate_0 1000 2 ) > $O/persist_ablate_order.txt 2>&1 || { tail $O/persist_ablate_order.txt; exit 1; }
grep -A1 "N= 20" $O/persist_ablate_order.txt
timeout -k 10 300 tools/ptile_sc1_stress 4096 > $O/ptile_sc1_stress.txt 2>&1; echo "stress rc=$?"; cat $O/ptile_sc1_stress.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_slab.py tests/test_gpu_ptile.py -x -q > $O/call6_pytest.txt 2>&1 || { tail -40 $O/call6_pytest.txt; exit 1; }
tail -3 $O/call6_pytest.txt
for w in mf_n500_b1000 langevin_n500_b1000 dl_n1000_b32 dl_n1000_b1000; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail -20 $O/bench_$w.err; exit 1; }
done
echo CALL6_DONE
