#!/bin/bash
# round 6, GPU call 13: rocprofv3 kernel statistics of every bench workload, TTS of all solvers
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
PART=stats timeout -k 10 900 bash tools/profile_round.sh r06 > $O/profile_stats.txt 2>&1 || { tail -5 $O/profile_stats.txt; ls gpurun_out/prof | tail -3; exit 1; }
tail -2 $O/profile_stats.txt
timeout -k 10 300 python tools/tts_report.py --out $O/tts.json > $O/tts.txt 2> $O/tts.err || { tail -20 $O/tts.err; exit 1; }
cat $O/tts.txt
echo CALL13_DONE
