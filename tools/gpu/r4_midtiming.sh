cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_midtiming; rm -rf $O; mkdir -p $O
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py > $O/out.log 2> $O/err.log; echo "rc=$?"
N=$(grep -n "wg-trace\] block1a" $O/err.log | tail -1 | cut -d: -f1)
tail -n +$N $O/err.log | grep "wg-trace\|wg-phase\|front-timing\|mid-timing\|chain-timing" | cut -c1-330
