#!/bin/bash
# round 6, GPU call 3: the row-owner kernel's variant sweep (policy data) + first lines of the new bench protocol
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python bench.py --workload dl_n20_b1000 > $O/bench_dl_n20_b1000.json 2> $O/bench_dl_n20_b1000.err || { tail -20 $O/bench_dl_n20_b1000.err; exit 1; }
timeout -k 10 300 python bench.py --workload dl_n20_b100 > $O/bench_dl_n20_b100.json 2> $O/bench_dl_n20_b100.err || { tail -20 $O/bench_dl_n20_b100.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err || { tail -20 $O/bench_steps20.err; exit 1; }
echo BENCH_DONE
timeout -k 10 1100 python tools/persist_policy_sweep.py > $O/persist_policy.jsonl 2> $O/persist_policy.err || { tail -20 $O/persist_policy.err; exit 1; }
echo CALL3_DONE
