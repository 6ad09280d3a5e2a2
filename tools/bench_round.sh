#!/bin/bash
# The round's un-profiled bench lines on the GPU box (developer tool): gpurun_out/<tag>_bench*.json
#   usage: tools/bench_round.sh <round tag, e.g. r03>      (then tools/collect_profiles.sh <tag>)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_like.json 2>> $O/${TAG}_bench.err || exit 1
python3 bench.py --workload pl_n2000_b512 --post adam > $O/${TAG}_bench_pl_adam.json 2>> $O/${TAG}_bench.err || exit 1
for w in dl_n20_b100 dl_n20_b1000 mf_n20_b1000 langevin_n20_b1000 pl_n20_b1000 dl_n70_b1000 dl_n160_b1000 mf_n257_b1000 dl_n100_b1000 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 langevin_n300_b1000 dl_n300_b1000 langevin_n640_b512 dl_n640_b512 dl_n1000_b256 langevin_n1000_b256 langevin_n1000_b1000 mf_n1000_b1000 dl_n1000_b2000 dl_n1000_b4000 pl_n2000_b1024; do
  python3 bench.py --workload $w > $O/${TAG}_bench_$w.json 2>> $O/${TAG}_bench.err || exit 1
done
# small batches (column-slab kernel): 8192 steps = two launches
for w in dl_n1000_b1 dl_n1000_b8 dl_n1000_b32 langevin_n1000_b32 mf_n500_b32 pl_n2000_b32; do
  python3 bench.py --workload $w --steps 8192 --warmup 1024 > $O/${TAG}_bench_$w.json 2>> $O/${TAG}_bench.err || exit 1
done
CCVM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 1000 --warmup 100 --no-cpu-baseline > $O/${TAG}_bench_gpus2_share.json 2>> $O/${TAG}_bench.err || exit 1
# strong scaling (BASELINE config 5's global batch split over the ranks), rehearsed on one GPU over gloo
CCVM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --workload pl_n2000_b512 --post adam --global-batch 4096 --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_gpus2_share_strong.json 2>> $O/${TAG}_bench.err || exit 1
# the driver's launch line with FIVE ranks on the one GPU (launcher + 5 = the six processes the pool allows on a card)
CCVM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 5 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_gpus5_share.json 2>> $O/${TAG}_bench.err || exit 1
echo BENCH_DONE
