#!/bin/bash
# Copy the summaries of tools/profile_round.sh (+ the bench lines in gpurun_out/) into profiles/ (developer tool).
#   usage: tools/collect_profiles.sh r02
TAG=${1:-r02}; P=gpurun_out/prof
for w in pl_n2000_b512 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 dl_n100_b1000; do cp $P/${w}_stats/s_kernel_stats.csv profiles/${TAG}_${w}_kernel_stats.csv; done
cp $P/dl_n1000_b1000_stats/s_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
for n in 320 384 448 500; do for k in auto nocluster; do cp $P/lv_n${n}_${k}_stats/s_kernel_stats.csv profiles/${TAG}_mid_n${n}_b1000_${k}_kernel_stats.csv; done; done
CMD='rocprofv3 --pmc <counter set> --kernel-trace --output-format csv -- python3 bench.py --workload W --steps 40 --warmup 5 [N = 500 workloads: --steps 1000 --warmup 200] --spinup-ms 0 --no-cpu-baseline (one pass per counter set: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*; tools/profile_round.sh)'
python tools/pmc_summary.py --kernel 'step_kernel<0, false, 0, 1' --name 'ccvm::step_kernel<0, false, 0, 1, false, 0> = DL step, N=1000, B=1000 (headline)' --command "${CMD/W/dl_n1000_b1000}" --out profiles/${TAG}_bench_pmc.json $P/dl_n1000_b1000_pmc_fetch $P/dl_n1000_b1000_pmc_write $P/dl_n1000_b1000_pmc_sq1 $P/dl_n1000_b1000_pmc_tcc > /dev/null
python tools/pmc_summary.py --kernel 'cluster_kernel<2' --steps-per-dispatch 600 --name 'ccvm::cluster_kernel<2, false, 4, false> = Langevin, N=500, B=1000 (one launch per chunk of steps: 200 and 1000 steps here)' --command "${CMD/W/langevin_n500_b1000}" --out profiles/${TAG}_langevin_n500_b1000_pmc.json $P/langevin_n500_b1000_pmc_fetch $P/langevin_n500_b1000_pmc_write $P/langevin_n500_b1000_pmc_sq1 $P/langevin_n500_b1000_pmc_tcc > /dev/null
python tools/pmc_summary.py --kernel 'cluster_kernel<0' --steps-per-dispatch 600 --name 'ccvm::cluster_kernel<0, false, 4, false> = DL, N=500, B=1000 (one launch per chunk of steps: 200 and 1000 steps here)' --command "${CMD/W/dl_n500_b1000}" --out profiles/${TAG}_dl_n500_b1000_pmc.json $P/dl_n500_b1000_pmc_fetch $P/dl_n500_b1000_pmc_write $P/dl_n500_b1000_pmc_sq1 $P/dl_n500_b1000_pmc_tcc > /dev/null
for n in 320 384 448 500; do for k in auto nocluster; do cp $P/lv_n${n}_${k}.txt profiles/${TAG}_mid_n${n}_b1000_${k}_unprofiled_timing.txt; done; done
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_bench_driver_like.json profiles/${TAG}_bench_steps20.json
cp gpurun_out/${TAG}_bench_gpus2_share.json profiles/${TAG}_bench_gpus2_share_rehearsal.json
cp gpurun_out/${TAG}_bench_pl_adam.json profiles/${TAG}_bench_pl_n2000_b512_adam.json
for w in dl_n100_b1000 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000; do cp gpurun_out/${TAG}_bench_$w.json profiles/${TAG}_bench_$w.json; done
