#!/bin/bash
# Copy the summaries of tools/profile_round.sh (+ the bench lines of tools/bench_round.sh in gpurun_out/) into profiles/
# (developer tool).    usage: tools/collect_profiles.sh r06
TAG=${1:-r06}; P=gpurun_out/prof
STATS="dl_n160_b1000 mf_n257_b1000 dl_n300_b1000 langevin_n300_b1000 pl_n2000_b512 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 dl_n100_b1000 dl_n1000_b256 langevin_n1000_b256 dl_n1000_b1 dl_n1000_b8 dl_n1000_b32 langevin_n1000_b32 mf_n500_b32 pl_n2000_b32 dl_n20_b100 dl_n20_b1000 mf_n20_b1000 langevin_n20_b1000 dl_n70_b1000"
for w in $STATS; do [ -f $P/${w}_stats/s_kernel_stats.csv ] && cp $P/${w}_stats/s_kernel_stats.csv profiles/${TAG}_${w}_kernel_stats.csv; done
cp $P/dl_n1000_b1000_stats/s_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
CMD='rocprofv3 --pmc <counter set> --kernel-trace --output-format csv -- python3 bench.py --workload W --steps 40 --warmup 5 [persistent kernels: --steps 1000 --warmup 200] --repeats 1 --spinup-ms 0 --no-cpu-baseline (one pass per counter set: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*; tools/profile_round.sh)'
pmc() {  # workload, output stem, kernel substring, steps per dispatch, description
  local w=$1 stem=$2 k=$3 spd=$4 name=$5
  [ -d $P/${w}_pmc_sq1 ] || return 0
  python tools/pmc_summary.py --kernel "$k" --steps-per-dispatch $spd --name "$name" --command "${CMD/W/$w}" --out profiles/${TAG}_${stem}_pmc.json $P/${w}_pmc_fetch $P/${w}_pmc_write $P/${w}_pmc_sq1 $P/${w}_pmc_tcc > /dev/null
}
PERS='(one launch per chunk of steps: 200 and 1000 steps here)'
pmc dl_n1000_b1000 bench 'ptile_kernel<0' 600 "ccvm::ptile_kernel<0> = DL, N=1000, B=1000 (headline; $PERS)"
pmc pl_n2000_b512 pl_n2000_b512 'ptile_kernel<2' 600 "ccvm::ptile_kernel<2> = pumped Langevin, N=2000, B=512 $PERS"
pmc langevin_n500_b1000 langevin_n500_b1000 'cluster_kernel<2' 600 "ccvm::cluster_kernel<2, false, 4, false> = Langevin, N=500, B=1000 $PERS"
pmc mf_n500_b1000 mf_n500_b1000 'cluster_kernel<1' 600 "ccvm::cluster_kernel<1, false, 4, false> = MF, N=500, B=1000 $PERS"
pmc dl_n100_b1000 dl_n100_b1000 'persist_kernel<0, false, 64, 2, 7, 4, 2, 1>' 600 "ccvm::persist_kernel<0, false, 64, 2, 7, 4, 2, 1> = DL, N=100, B=1000: K split + noise producer waves $PERS"
pmc dl_n70_b1000 dl_n70_b1000 'persist_kernel<0, false, 32, 3, 5, 4, 1, 1' 600 "ccvm::persist_kernel<0, false, 32, 3, 5, 4, 1, 1> = DL, N=70, B=1000: three 32-column waves of eight rows + noise producer waves $PERS"
pmc dl_n160_b1000 dl_n160_b1000 'persist_kernel<0, false, 64, 3, 10, 4, 2, 0, 2' 600 "ccvm::persist_kernel<0, false, 64, 3, 10, 4, 2, 0, 2, 0> = DL, N=160, B=1000: three waves side by side x two K halves, two row sets per twelve-wave workgroup $PERS"
pmc mf_n257_b1000 mf_n257_b1000 'persist_kernel<1, false, 64, 5, 17' 600 "ccvm::persist_kernel<1, false, 64, 5, 17, 4, 2, 0, 0, 56, 80> = MF, N=257, B=1000: five waves side by side, K split 80 | 192, the long parts' last 112 fragments in LDS $PERS"
pmc dl_n300_b1000 dl_n300_b1000 'persist_kernel<0, false, 64, 5, 19' 600 "ccvm::persist_kernel<0, false, 64, 5, 19, 4, 2, 0, 0, 48, 104> = DL, N=300, B=1000: five waves side by side, K split 104 | 200, the long parts' last 96 fragments in LDS $PERS"
pmc langevin_n300_b1000 langevin_n300_b1000 'persist_kernel<2, false, 64, 5, 19' 600 "ccvm::persist_kernel<2, false, 64, 5, 19, 4, 2, 0, 0, 48, 104> = Langevin, N=300, B=1000: the same $PERS"
pmc dl_n20_b1000 dl_n20_b1000 'persist_kernel<0, false, 32, 1, 2, 2, 1, 1>' 600 "ccvm::persist_kernel<0, false, 32, 1, 2, 2, 1, 1> = DL, N=20 (tuningH020-100-0), B=1000: one wave per row set + noise producer waves $PERS"
pmc dl_n20_b100 dl_n20_b100 'persist_kernel<0, false, 32, 1, 2, 2, 1, 1>' 600 "ccvm::persist_kernel<0, false, 32, 1, 2, 2, 1, 1> = DL, N=20 (test020-100-10), B=100: BASELINE config 1 $PERS"
pmc mf_n20_b1000 mf_n20_b1000 'persist_kernel<1, false, 32, 1, 2, 2, 1, 1>' 600 "ccvm::persist_kernel<1, false, 32, 1, 2, 2, 1, 1> = MF, N=20, B=1000 $PERS"
pmc langevin_n20_b1000 langevin_n20_b1000 'persist_kernel<2, false, 32, 1, 2, 2, 1, 1>' 600 "ccvm::persist_kernel<2, false, 32, 1, 2, 2, 1, 1> = Langevin, N=20, B=1000 $PERS"
pmc dl_n1000_b32 dl_n1000_b32 'slab_kernel<0, 8, 128' 600 "ccvm::slab_kernel<0, 8, 128> = DL, N=1000, B=32 $PERS"
pmc dl_n1000_b256 dl_n1000_b256 'step_kernel<0, false, 0, 4' 1 'ccvm::step_kernel<0, false, 0, 4, false, 0> = DL step, N=1000, B=256 (32 x 32 split-K tiles)'
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_bench_driver_like.json profiles/${TAG}_bench_steps20.json
tail -n 1 gpurun_out/${TAG}_bench_gpus2_share.json > profiles/${TAG}_bench_gpus2_share_rehearsal.json
tail -n 1 gpurun_out/${TAG}_bench_gpus2_share_strong.json > profiles/${TAG}_bench_gpus2_share_strong_rehearsal.json
cp gpurun_out/${TAG}_bench_pl_adam.json profiles/${TAG}_bench_pl_n2000_b512_adam.json
for f in gpurun_out/${TAG}_bench_*_*.json; do
  w=${f#gpurun_out/${TAG}_bench_}; w=${w%.json}
  case $w in driver_like|pl_adam|gpus2_share|gpus2_share_strong|gpus5_share) continue;; esac
  cp $f profiles/${TAG}_bench_$w.json
done
for how in launcher torch_distributed_run; do
  [ -f gpurun_out/bench_gpus4_share_rehearsal_$how.json ] && cp gpurun_out/bench_gpus4_share_rehearsal_$how.json profiles/${TAG}_bench_gpus4_share_rehearsal_$how.json
done
[ -f gpurun_out/${TAG}_bench_gpus5_share.json ] && tail -n 1 gpurun_out/${TAG}_bench_gpus5_share.json > profiles/${TAG}_bench_gpus5_share_rehearsal.json
