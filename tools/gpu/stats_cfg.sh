# rocprofv3 kernel stats of the fine-tune and streaming bench commands (supporting evidence for BASELINE configs[3], [4])
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in finetune stream; do
  O=$GRAFT_REPO_ROOT/gpurun_out/stats_$c; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $O/log.txt 2>&1 )
  f=$(find $O -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/${R:-r03}_kernel_stats_$c.csv; head -8 $f | cut -c1-150
done
