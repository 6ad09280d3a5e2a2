#!/bin/bash
# Round profiles on the GPU box (developer tool): rocprofv3 kernel stats of every bench workload and the PMC
# passes of the headline and the N = 500 kernels.  Output: gpurun_out/prof/<tag>/...  (copy summaries to profiles/).
#   usage: tools/profile_round.sh <round tag, e.g. r02>
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
if [ "${PMC_ONLY:-0}" = 0 ]; then
for w in dl_n1000_b1000 pl_n2000_b512 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 dl_n100_b1000; do
  rocprofv3 --kernel-trace --stats -d $OUT/${w}_stats -o s --output-format csv -- python3 bench.py --workload $w --steps 3000 --warmup 500 --no-cpu-baseline > $OUT/${w}_stats.json 2> $OUT/${w}_stats.err || exit 1
  echo "stats $w done"
done
# mid-size regime: cluster kernel vs per-step tile kernel
for n in 320 384 448 500; do
  for k in auto nocluster; do
    if [ $k = auto ]; then unset CCVM_AMD_KERNEL; else export CCVM_AMD_KERNEL=nocluster; fi
    # timing WITHOUT the profiler (its per-dispatch overhead inflates a per-step kernel by 2-6 us and leaves a
    # persistent kernel untouched), then the same run under the profiler for the kernel names and launch counts
    python3 tools/time_small.py langevin:$n:1000 mf:$n:1000 dl:$n:1000 > $OUT/lv_n${n}_${k}.txt 2>&1 || exit 1
    rocprofv3 --kernel-trace --stats -d $OUT/lv_n${n}_${k}_stats -o s --output-format csv -- python3 tools/time_small.py langevin:$n:1000 mf:$n:1000 > $OUT/lv_n${n}_${k}_profiled.txt 2>&1 || exit 1
  done
  unset CCVM_AMD_KERNEL
  echo "mid $n done"
done
fi
PMC_ONLY=${PMC_ONLY:-0}
pmc() {  # name, workload, counters...   (one launch per step: 45 dispatches; persistent kernels: 200 + 1000 steps in two)
  local name=$1 w=$2; shift 2
  local steps="--steps 40 --warmup 5"
  case $w in *_n500_*) steps="--steps 1000 --warmup 200";; esac
  rocprofv3 --pmc "$@" --kernel-trace -d $OUT/${w}_pmc_$name -o pmc --output-format csv -- python3 bench.py --workload $w $steps --spinup-ms 0 --no-cpu-baseline > $OUT/${w}_pmc_$name.log 2>&1 || exit 1
  echo "pmc $w $name done"
}
for w in dl_n1000_b1000 langevin_n500_b1000 dl_n500_b1000; do
  pmc fetch $w FETCH_SIZE
  pmc write $w WRITE_SIZE
  pmc sq1 $w SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
  pmc tcc $w TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
echo ALL_DONE
