#!/bin/bash
# Copy the summaries of tools/profile_round.sh (+ the bench lines in gpurun_out/) into profiles/ (developer tool).
#   usage: tools/collect_profiles.sh r03
TAG=${1:-r03}; P=gpurun_out/prof
for w in pl_n2000_b512 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 dl_n100_b1000 dl_n1000_b256 langevin_n1000_b256 dl_n1000_b1 dl_n1000_b8 dl_n1000_b32 langevin_n1000_b32 mf_n500_b32 pl_n2000_b32; do cp $P/${w}_stats/s_kernel_stats.csv profiles/${TAG}_${w}_kernel_stats.csv; done
cp $P/dl_n1000_b1000_stats/s_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
CMD='rocprofv3 --pmc <counter set> --kernel-trace --output-format csv -- python3 bench.py --workload W --steps 40 --warmup 5 [persistent kernels: --steps 1000 --warmup 200] --spinup-ms 0 --no-cpu-baseline (one pass per counter set: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*; tools/profile_round.sh)'
python tools/pmc_summary.py --kernel 'step_kernel<0, false, 0, 1' --name 'ccvm::step_kernel<0, false, 0, 1, false, 0> = DL step, N=1000, B=1000 (headline)' --command "${CMD/W/dl_n1000_b1000}" --out profiles/${TAG}_bench_pmc.json $P/dl_n1000_b1000_pmc_fetch $P/dl_n1000_b1000_pmc_write $P/dl_n1000_b1000_pmc_sq1 $P/dl_n1000_b1000_pmc_tcc > /dev/null
python tools/pmc_summary.py --kernel 'cluster_kernel<2' --steps-per-dispatch 600 --name 'ccvm::cluster_kernel<2, false, 4, false> = Langevin, N=500, B=1000 (one launch per chunk of steps: 200 and 1000 steps here)' --command "${CMD/W/langevin_n500_b1000}" --out profiles/${TAG}_langevin_n500_b1000_pmc.json $P/langevin_n500_b1000_pmc_fetch $P/langevin_n500_b1000_pmc_write $P/langevin_n500_b1000_pmc_sq1 $P/langevin_n500_b1000_pmc_tcc > /dev/null
python tools/pmc_summary.py --kernel 'persist_kernel<0, false, 64, 2, 7, 4, 2>' --steps-per-dispatch 600 --name 'ccvm::persist_kernel<0, false, 64, 2, 7, 4, 2> = DL, N=100, B=1000 (one launch per chunk of steps: 200 and 1000 steps here)' --command "${CMD/W/dl_n100_b1000}" --out profiles/${TAG}_dl_n100_b1000_pmc.json $P/dl_n100_b1000_pmc_fetch $P/dl_n100_b1000_pmc_write $P/dl_n100_b1000_pmc_sq1 $P/dl_n100_b1000_pmc_tcc > /dev/null
python tools/pmc_summary.py --kernel 'slab_kernel<0, 8, 128' --steps-per-dispatch 600 --name 'ccvm::slab_kernel<0, 8, 128> = DL, N=1000, B=32 (one launch per chunk of steps: 200 and 1000 steps here)' --command "${CMD/W/dl_n1000_b32}" --out profiles/${TAG}_dl_n1000_b32_pmc.json $P/dl_n1000_b32_pmc_fetch $P/dl_n1000_b32_pmc_write $P/dl_n1000_b32_pmc_sq1 $P/dl_n1000_b32_pmc_tcc > /dev/null
python tools/pmc_summary.py --kernel 'step_kernel<0, false, 0, 4' --name 'ccvm::step_kernel<0, false, 0, 4, false, 0> = DL step, N=1000, B=256 (32 x 32 split-K tiles)' --command "${CMD/W/dl_n1000_b256}" --out profiles/${TAG}_dl_n1000_b256_pmc.json $P/dl_n1000_b256_pmc_fetch $P/dl_n1000_b256_pmc_write $P/dl_n1000_b256_pmc_sq1 $P/dl_n1000_b256_pmc_tcc > /dev/null
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_bench_driver_like.json profiles/${TAG}_bench_steps20.json
tail -n 1 gpurun_out/${TAG}_bench_gpus2_share.json > profiles/${TAG}_bench_gpus2_share_rehearsal.json
tail -n 1 gpurun_out/${TAG}_bench_gpus2_share_strong.json > profiles/${TAG}_bench_gpus2_share_strong_rehearsal.json
cp gpurun_out/${TAG}_bench_pl_adam.json profiles/${TAG}_bench_pl_n2000_b512_adam.json
for w in dl_n100_b1000 mf_n500_b1000 langevin_n500_b1000 dl_n500_b1000 dl_n1000_b256 langevin_n1000_b256 dl_n1000_b1 dl_n1000_b8 dl_n1000_b32 langevin_n1000_b32 mf_n500_b32 pl_n2000_b32; do cp gpurun_out/${TAG}_bench_$w.json profiles/${TAG}_bench_$w.json; done
cp gpurun_out/size_sweep_auto.txt profiles/${TAG}_size_sweep_auto.txt; cp gpurun_out/size_sweep_tile.txt profiles/${TAG}_size_sweep_tile.txt
python tools/make_size_sweep_md.py $TAG > profiles/${TAG}_size_sweep.md
cp gpurun_out/small_batch_sweep.jsonl profiles/${TAG}_small_batch_sweep.jsonl
python tools/make_small_batch_md.py $TAG > profiles/${TAG}_small_batch.md
cp gpurun_out/regime_map.jsonl profiles/${TAG}_regime_map.jsonl; python tools/regime_map.py --md $TAG > profiles/${TAG}_regime_map.md
