cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_lanemodes2; rm -rf $O; mkdir -p $O
for rep in 1 2; do
MODES=eager,join LANES=1 BIG=0 timeout 300 python tools/lane_modes.py 1024 2>&1 | grep max_batch | tee -a $O/split.txt
MODES=eager,join LANES=2 BIG=1 timeout 300 python tools/lane_modes.py 512 2>&1 | grep max_batch | tee -a $O/split.txt
MODES=eager,join LANES=4 BIG=1 timeout 300 python tools/lane_modes.py 256 2>&1 | grep max_batch | tee -a $O/split.txt
done
MODES=eager,join LANES=3 BIG=1 timeout 300 python tools/lane_modes.py 344 2>&1 | grep max_batch | tee -a $O/split.txt
MODES=eager,join LANES=2 BIG=1 timeout 300 python tools/lane_modes.py 1024 2>&1 | grep max_batch | tee -a $O/split.txt
MODES=eager,join LANES=4 BIG=1 timeout 300 python tools/lane_modes.py 512 2>&1 | grep max_batch | tee -a $O/split.txt
