#!/bin/bash
# Same-box A/B (developer tool): the slab kernel with the K values added in round 5 (384, 640, 896, 1152, 1792) against the
# previous library (tools/lib_prev.so = the build before; K in {512, 768, 1024, 1280, 1536, 2048}).  Writes gpurun_out/ab_slab_k.txt.
CASES="dl:300:8 dl:300:32 dl:300:128 dl:300:256 langevin:300:32 langevin:300:256 mf:350:64 dl:600:16 dl:600:128 langevin:640:64 mf:800:32 dl:896:64 langevin:1100:32 dl:1150:16 pl:1700:16 dl:1792:8"
{
echo "== previous library"; CCVM_AMD_LIB=$PWD/tools/lib_prev.so python3 tools/time_small.py $CASES
echo "== this library"; python3 tools/time_small.py $CASES
echo "== previous library again"; CCVM_AMD_LIB=$PWD/tools/lib_prev.so python3 tools/time_small.py $CASES
} > gpurun_out/ab_slab_k.txt 2>&1
