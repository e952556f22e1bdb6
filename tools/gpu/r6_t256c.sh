cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_t256c; rm -rf $O; mkdir -p $O
AB_OPTION=block_tiles AB_VALUES=2,1 timeout 300 python tools/kernel_table.py 256 20 > $O/table_256.txt 2>&1; awk '/pass 2/{f=1} f' $O/table_256.txt
