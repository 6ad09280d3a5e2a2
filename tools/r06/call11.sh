#!/bin/bash
# round 6, GPU call 11: the whole GPU suite on the round's code
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
timeout -k 10 1150 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/call11_pytest.txt 2>&1; rc=$?
tail -15 $O/call11_pytest.txt | cut -c1-250
exit $rc
