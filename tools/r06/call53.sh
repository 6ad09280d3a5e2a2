#!/bin/bash
# round 6, call 53: kernel statistics, counter passes and bench lines of the five-wave workloads again (the unequal K split)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; P=$R/gpurun_out/prof; mkdir -p $O $P
for w in dl_n300_b1000 langevin_n300_b1000 mf_n257_b1000; do
  rm -rf $P/${w}_stats $P/${w}_pmc_*
  rocprofv3 --kernel-trace --stats -d $P/${w}_stats -o s --output-format csv -- python3 bench.py --workload $w --steps 3000 --warmup 500 --no-cpu-baseline > $P/${w}_stats.json 2> $P/${w}_stats.err || { echo STATS_FAILED $w; exit 1; }
  cp $P/${w}_stats/s_kernel_stats.csv $O/${w}_kernel_stats.csv
  pmc() { local name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $P/${w}_pmc_$name -o pmc --output-format csv -- python3 bench.py --workload $w --steps 1000 --warmup 200 --repeats 1 --spinup-ms 0 --no-cpu-baseline > $P/${w}_pmc_$name.log 2>&1 || { echo PMC_FAILED $w $name; exit 1; }; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
  pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  case $w in
    dl_n300_b1000) k='persist_kernel<0, false, 64, 5, 19'; what="ccvm::persist_kernel<0, false, 64, 5, 19, 4, 2, 0, 0, 48, 104> = DL, N=300, B=1000: five waves side by side, K split 104 | 200, the long parts' last 96 fragments in LDS";;
    langevin_n300_b1000) k='persist_kernel<2, false, 64, 5, 19'; what="ccvm::persist_kernel<2, false, 64, 5, 19, 4, 2, 0, 0, 48, 104> = Langevin, N=300, B=1000: the same";;
    *) k='persist_kernel<1, false, 64, 5, 17'; what="ccvm::persist_kernel<1, false, 64, 5, 17, 4, 2, 0, 0, 56, 80> = MF, N=257, B=1000: five waves side by side, K split 80 | 192, the long parts' last 112 fragments in LDS";;
  esac
  python tools/pmc_summary.py --kernel "$k" --steps-per-dispatch 600 --name "$what (one launch per chunk of steps: 200 and 1000 steps here)" --command "rocprofv3 --pmc <counter set> --kernel-trace --output-format csv -- python3 bench.py --workload $w --steps 1000 --warmup 200 --repeats 1 --spinup-ms 0 --no-cpu-baseline (one pass per counter set: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*)" --out $O/${w}_pmc.json $P/${w}_pmc_fetch $P/${w}_pmc_write $P/${w}_pmc_sq1 $P/${w}_pmc_tcc > /dev/null || echo PMC_SUMMARY_FAILED $w
  cp $O/${w}_pmc.json profiles/r06_${w}_pmc.json
  timeout -k 10 200 python bench.py --workload $w > gpurun_out/r06_bench_$w.json 2> $O/bench_$w.err || { echo BENCH_FAILED $w; tail -5 $O/bench_$w.err; exit 1; }
  echo "$w done"
done
echo CALL53_DONE
