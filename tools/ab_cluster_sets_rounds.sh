CASES="langevin:640:1500 langevin:576:1500 langevin:640:1200 langevin:768:1100 langevin:768:1500 mf:640:1500 dl:640:1500 langevin:640:3000 mf:704:1400 langevin:576:1100"
echo "== three row sets"; CCVM_AMD_KERNEL=cluster CCVM_AMD_CLUSTER_SETS=3 python3 tools/time_small.py $CASES
echo "== two row sets"; CCVM_AMD_KERNEL=cluster CCVM_AMD_CLUSTER_SETS=2 python3 tools/time_small.py $CASES
echo "== tiles"; CCVM_AMD_KERNEL=nocluster python3 tools/time_small.py $CASES
echo "== default"; python3 tools/time_small.py $CASES
