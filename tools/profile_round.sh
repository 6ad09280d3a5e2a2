#!/bin/bash
# Round profiles on the GPU box (developer tool): rocprofv3 kernel stats of every bench workload and the PMC
# passes of the headline, the N = 500, the N = 100 and the small-batch kernels.  Output: gpurun_out/prof/...
# (copy the summaries to profiles/ with tools/collect_profiles.sh).
#   usage: PART=stats|pmc tools/profile_round.sh <round tag, e.g. r03>
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
if [ "${PART:-stats}" = stats ]; then
for w in dl_n1000_b1000 pl_n2000_b512 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 dl_n100_b1000 dl_n1000_b256 langevin_n1000_b256 dl_n20_b100 dl_n20_b1000 mf_n20_b1000 langevin_n20_b1000 dl_n70_b1000 dl_n300_b1000 langevin_n300_b1000 dl_n160_b1000 mf_n257_b1000; do
  rocprofv3 --kernel-trace --stats -d $OUT/${w}_stats -o s --output-format csv -- python3 bench.py --workload $w --steps 3000 --warmup 500 --no-cpu-baseline > $OUT/${w}_stats.json 2> $OUT/${w}_stats.err || exit 1
  echo "stats $w done"
done
for w in dl_n1000_b1 dl_n1000_b8 dl_n1000_b32 langevin_n1000_b32 mf_n500_b32 pl_n2000_b32; do
  rocprofv3 --kernel-trace --stats -d $OUT/${w}_stats -o s --output-format csv -- python3 bench.py --workload $w --steps 8192 --warmup 1024 --no-cpu-baseline > $OUT/${w}_stats.json 2> $OUT/${w}_stats.err || exit 1
  echo "stats $w done"
done
exit 0
fi
pmc() {  # name, workload, counters...   (one launch per step: 45 dispatches; persistent kernels: 200 + 1000 steps in two)
  local name=$1 w=$2; shift 2
  local steps="--steps 40 --warmup 5"
  case $w in *_n500_*|*_n300_*|*_n160_*|*_n257_*|*_n100_*|*_n20_*|*_n70_*|*_b32|*_b8|*_b1|dl_n1000_b1000|pl_n2000_b512) steps="--steps 1000 --warmup 200";; esac  # persistent kernels
  rocprofv3 --pmc "$@" --kernel-trace -d $OUT/${w}_pmc_$name -o pmc --output-format csv -- python3 bench.py --workload $w $steps --repeats 1 --spinup-ms 0 --no-cpu-baseline > $OUT/${w}_pmc_$name.log 2>&1 || exit 1
  echo "pmc $w $name done"
}
if [ "${PART}" = classes ]; then
  # executed instruction classes of the issue-bound row-owner kernel and of the headline kernel
  for w in dl_n100_b1000 dl_n1000_b1000; do
    pmc cls1 $w SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH
    pmc cls2 $w SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32
  done
  echo ALL_DONE; exit 0
fi
for w in dl_n1000_b1000 pl_n2000_b512 langevin_n500_b1000 mf_n500_b1000 dl_n100_b1000 dl_n1000_b32 dl_n1000_b256 dl_n20_b100 dl_n20_b1000 mf_n20_b1000 langevin_n20_b1000 dl_n70_b1000 dl_n300_b1000 langevin_n300_b1000 dl_n160_b1000 mf_n257_b1000; do
  pmc fetch $w FETCH_SIZE
  pmc write $w WRITE_SIZE
  pmc sq1 $w SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
  pmc tcc $w TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
echo ALL_DONE
