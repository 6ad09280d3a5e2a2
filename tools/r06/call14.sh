#!/bin/bash
# round 6, GPU call 14: PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*: a pass each) of the BASELINE shapes and the N = 20 / 70 workloads
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
PART=pmc timeout -k 10 1150 bash tools/profile_round.sh r06 > $O/profile_pmc.txt 2>&1 || { tail -5 $O/profile_pmc.txt; exit 1; }
tail -3 $O/profile_pmc.txt
echo CALL14_DONE
