#!/bin/bash
# round 6, GPU call 15: the row-owner kernel's variant sweep on the final kernel (schedule-row relay in), bench lines with models
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python tools/persist_policy_sweep.py > $O/persist_policy2.jsonl 2> $O/persist_policy2.err || { tail -20 $O/persist_policy2.err; exit 1; }
wc -l $O/persist_policy2.jsonl
python3 bench.py > gpurun_out/r06_bench.json 2>> gpurun_out/r06_bench.err || exit 1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_like.json 2>> gpurun_out/r06_bench.err || exit 1
for w in dl_n20_b100 dl_n20_b1000 mf_n20_b1000 langevin_n20_b1000 pl_n20_b1000 dl_n70_b1000 dl_n100_b1000; do
  python3 bench.py --workload $w > gpurun_out/r06_bench_$w.json 2>> gpurun_out/r06_bench.err || exit 1
done
echo CALL15_DONE
