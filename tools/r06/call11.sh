#!/bin/bash
# round 6, GPU call 11: the whole GPU suite on the round's code
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
timeout -k 10 1150 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" > $O/call11_pytest.txt; rc=${PIPESTATUS[0]}
tail -5 $O/call11_pytest.txt; grep -E "faulty launch" $O/call11_pytest.txt | head -8
exit $rc
