#!/bin/bash
# round 6, GPU call 12: the rest of the GPU suite, then the round's bench lines
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_slab.py tests/test_gpu_streams.py tests/test_gpu_thick_goldens.py tests/test_gpu_tile_shapes.py -q -x -p no:cacheprovider > $O/call12_pytest.txt 2>&1; rc=$?
tail -4 $O/call12_pytest.txt | cut -c1-250
[ $rc = 0 ] || exit $rc
timeout -k 10 900 bash tools/bench_round.sh r06 > $O/bench_round.txt 2>&1 || { tail -5 $O/bench_round.txt; tail -20 gpurun_out/r06_bench.err; exit 1; }
tail -2 $O/bench_round.txt
echo CALL12_DONE
