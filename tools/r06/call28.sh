#!/bin/bash
# round 6, call 28: the narrow shape (64 < N <= 96) where it pays -- the whole GPU suite; the N = 70 workload's bench line,
# kernel statistics and counter passes again (its kernel changed); the cells of the rule under the default policy
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; P=$R/gpurun_out/prof; mkdir -p $O $P
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_cw32.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_cw32.txt; exit 1; }
tail -2 $O/gpu_suite_cw32.txt
w=dl_n70_b1000
timeout -k 10 200 python bench.py --workload $w > gpurun_out/r06_bench_$w.json 2> $O/bench_$w.err || { echo BENCH_FAILED; tail -5 $O/bench_$w.err; exit 1; }
rm -rf $P/${w}_stats $P/${w}_pmc_*
rocprofv3 --kernel-trace --stats -d $P/${w}_stats -o s --output-format csv -- python3 bench.py --workload $w --steps 3000 --warmup 500 --no-cpu-baseline > $P/${w}_stats.json 2> $P/${w}_stats.err || { echo STATS_FAILED; exit 1; }
pmc() { local name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $P/${w}_pmc_$name -o pmc --output-format csv -- python3 bench.py --workload $w --steps 1000 --warmup 200 --repeats 1 --spinup-ms 0 --no-cpu-baseline > $P/${w}_pmc_$name.log 2>&1 || { echo PMC_FAILED $name; exit 1; }; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# (the counter passes are read on the box: only their summary travels back)
python tools/pmc_summary.py --kernel 'persist_kernel<0, false, 32, 3, 5, 4, 1, 1' --steps-per-dispatch 600 --name "ccvm::persist_kernel<0, false, 32, 3, 5, 4, 1, 1> = DL, N=70, B=1000: three 32-column waves of eight rows + noise producer waves (one launch per chunk of steps: 200 and 1000 steps here)" --command "rocprofv3 --pmc <counter set> --kernel-trace --output-format csv -- python3 bench.py --workload $w --steps 1000 --warmup 200 --repeats 1 --spinup-ms 0 --no-cpu-baseline (one pass per counter set: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*)" --out $O/${w}_pmc.json $P/${w}_pmc_fetch $P/${w}_pmc_write $P/${w}_pmc_sq1 $P/${w}_pmc_tcc > /dev/null || echo PMC_SUMMARY_FAILED
cp $P/${w}_stats/s_kernel_stats.csv $O/${w}_kernel_stats.csv
CASES=""
for n in 65 70 80 96; do for b in 500 600 1000 1024 1100; do CASES="$CASES dl:$n:$b"; done; for b in 1000 1100 1500 2000 2100; do CASES="$CASES langevin:$n:$b"; done; done
for k in langevin mf; do for n in 70 96; do for b in 100 500 1000 1100; do CASES="$CASES $k:$n:$b:adam"; done; done; done
timeout -k 10 200 python3 tools/time_small.py $CASES > $O/persist_65_96_default.txt 2>&1 || { echo TIME_FAILED; tail -5 $O/persist_65_96_default.txt; exit 1; }
tail -2 $O/persist_65_96_default.txt
echo CALL28_DONE
