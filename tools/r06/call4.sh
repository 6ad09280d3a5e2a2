#!/bin/bash
# round 6, GPU call 4: clocks under light / heavy load, producers under the fitted policy, TTS of all solvers, thread sweep
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 120 tools/clock_probe > $O/clock_probe.txt 2>&1 || { tail $O/clock_probe.txt; exit 1; }
echo CLOCK_DONE
timeout -k 10 900 python -m pytest tests/test_gpu_persist_producers.py tests/test_gpu_persist_ksplit.py -x -q > $O/call4_pytest.txt 2>&1 || { tail -30 $O/call4_pytest.txt; exit 1; }
tail -2 $O/call4_pytest.txt
timeout -k 10 600 python tools/tts_report.py --out $O/tts.json > $O/tts.txt 2> $O/tts.err || { tail -20 $O/tts.err; exit 1; }
echo TTS_DONE
for w in dl_n100_b1000 mf_n20_b1000 langevin_n20_b1000 pl_n20_b1000 dl_n70_b1000; do
  timeout -k 10 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err || { tail -20 $O/bench_$w.err; exit 1; }
done
echo BENCH_DONE
timeout -k 10 600 python tools/cpu_thread_sweep.py dl_n1000_b1000 dl_n20_b1000 mf_n500_b1000 dl_n100_b1000 > $O/cpu_thread_sweep.md 2> $O/cpu_thread_sweep.err || { tail $O/cpu_thread_sweep.err; exit 1; }
echo CALL4_DONE
