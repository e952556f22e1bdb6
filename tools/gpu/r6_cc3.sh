cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_cc3; rm -rf $O; mkdir -p $O
for n in 1 2 3 4 6; do timeout 120 python tools/chain_concurrency_probe.py $n 200 2>&1 | tail -8 | tee -a $O/conc.txt; done
