# phase profile + ablations of mbconv_rows_kernel (timing build, built in the container)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_rows_timing; rm -rf $O; mkdir -p $O
export MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so
for ab in 0 1 4 8 16 32 12 60; do
  MKWS_ABLATE=$ab timeout 300 python tools/rows_timing.py 1024 2>&1 | grep "rows-timing" | tail -2 >> $O/timing.txt
done
cat $O/timing.txt
