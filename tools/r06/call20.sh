#!/bin/bash
# round 6, call 20: three waves side by side for 128 < N <= 192 -- parity of the variant build, then the A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
bash tools/ab_build.sh "-DCCVM_PERSIST_NCG3=1" > $O/ncg3_build.log 2>&1 || { echo BUILD_FAILED; tail -20 $O/ncg3_build.log; exit 1; }
CCVM_AMD_LIB=/tmp/libccvm_variant.so timeout -k 10 600 python -m pytest tests/test_gpu_persist_ksplit.py tests/test_gpu_persist_producers.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -q -m gpu -p no:cacheprovider > $O/ncg3_tests.txt 2>&1; echo "variant tests exit $?"; tail -15 $O/ncg3_tests.txt
timeout -k 10 500 bash tools/ab_persist_ncg3.sh > $O/ab_persist_ncg3.txt 2>&1 || { echo AB_FAILED; tail -5 $O/ab_persist_ncg3.txt; exit 1; }
tail -2 $O/ab_persist_ncg3.txt
echo CALL20_DONE
