cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_table256; rm -rf $O; mkdir -p $O
for B in 256 512; do timeout 300 python tools/kernel_table.py $B 20 > $O/table_$B.txt 2>&1; awk '/pass 1/{f=1} f' $O/table_$B.txt; done
