cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sh in "256000 32 16" "66560 96 24"; do
for v in 2 0; do
  O=$GRAFT_REPO_ROOT/gpurun_out/r5_k_$v; rm -rf $O; mkdir -p $O
  ( cd /tmp && MKWS_TRAIN_GEMM_TN2=$v timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/gemm_one.py $sh TN 20 > $O/log.txt 2>&1 )
  echo "== shape $sh tn2=$v"; f=$(find $O -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-160
done
done
