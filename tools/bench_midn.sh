set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O/prof
for w in ${MIDN:-langevin_n300_b1000 dl_n300_b1000 langevin_n640_b512 dl_n640_b512}; do
  python3 bench.py --workload $w > $O/r05_bench_$w.json 2>> $O/midn.err || exit 1
  rocprofv3 --kernel-trace --stats -d $O/prof/${w}_stats -o s --output-format csv -- python3 bench.py --workload $w --steps 3000 --warmup 500 --no-cpu-baseline > $O/prof/${w}_stats.json 2> $O/prof/${w}_stats.err || exit 1
done
python3 bench.py > $O/r05_bench_recheck.json 2>> $O/midn.err || exit 1
python3 bench.py --steps 20 --warmup 5 > $O/r05_bench_steps20_recheck.json 2>> $O/midn.err || exit 1
echo MIDN_DONE
