export CCVM_AMD_KERNEL=tile
for ks in 1 2 4; do echo "== KS=$ks"; CCVM_AMD_KS=$ks python3 tools/time_small.py dl:1700:1000 mf:1200:1000 dl:1800:1000 langevin:1700:1000 mf:1700:1000; done
