#!/bin/bash
# round 6, GPU call 8: schedule rows relayed by the producer waves; the whole GPU suite on the round's code
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 120 tools/persist_vs_abi 1000 > $O/persist_vs_abi3.txt 2>&1 || { tail $O/persist_vs_abi3.txt; exit 1; }
cat $O/persist_vs_abi3.txt | cut -c1-110
timeout -k 10 200 python tools/time_small.py dl:20:1000 dl:20:100 mf:20:1000 langevin:20:1000 dl:64:1000 dl:100:1000 langevin:100:1000 mf:100:1000 dl:128:1000 dl:200:1000 > $O/time_small_relay.txt 2>&1 || { tail $O/time_small_relay.txt; exit 1; }
cat $O/time_small_relay.txt | cut -c1-150
timeout -k 10 300 python -m pytest tests/test_gpu_persist_producers.py tests/test_gpu_api.py -x -q > $O/call8_pytest_first.txt 2>&1 || { tail -30 $O/call8_pytest_first.txt; exit 1; }
timeout -k 10 1050 python -m pytest tests -m gpu -x -q > $O/call8_pytest.txt 2>&1 || { tail -40 $O/call8_pytest.txt; exit 1; }
tail -3 $O/call8_pytest.txt
echo CALL8_DONE
