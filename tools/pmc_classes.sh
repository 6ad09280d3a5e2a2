#!/bin/bash
# Executed instruction classes of one bench workload (developer tool; GPU box): two rocprofv3 --pmc passes.
#   usage: tools/pmc_classes.sh <workload> [steps]     -> gpurun_out/prof/<workload>_pmc_cls{1,2}/
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; W=$1; S=${2:-1000}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace -d $OUT/${W}_pmc_cls1 -o pmc --output-format csv -- python3 bench.py --workload $W --steps $S --warmup 200 --spinup-ms 0 --no-cpu-baseline > $OUT/${W}_pmc_cls1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $OUT/${W}_pmc_cls2 -o pmc --output-format csv -- python3 bench.py --workload $W --steps $S --warmup 200 --spinup-ms 0 --no-cpu-baseline > $OUT/${W}_pmc_cls2.log 2>&1 || exit 1
echo done $W
