CASES="dl:2000:384 dl:1500:384 dl:768:2000 langevin:2000:384 dl:500:2000 dl:900:1500 mf:1500:128 dl:640:2000 langevin:900:2500 dl:900:800 dl:300:2000 mf:768:4000"
echo "== previous library (row-major runs)"; CCVM_AMD_LIB=$PWD/tools/lib_prev.so python tools/time_small.py $CASES 2>/dev/null
echo "== blocked order, default width (2 ks)"; python tools/time_small.py $CASES 2>/dev/null
for w in 2 4 8 16; do echo "== blocked order, width $w"; CCVM_AMD_XCD_XC=$w python tools/time_small.py $CASES 2>/dev/null; done
echo "== previous library again"; CCVM_AMD_LIB=$PWD/tools/lib_prev.so python tools/time_small.py $CASES 2>/dev/null
