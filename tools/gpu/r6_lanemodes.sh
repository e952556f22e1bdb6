cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_lanemodes; rm -rf $O; mkdir -p $O
LANES=1,2,4 BIG=0,1 timeout 400 python tools/lane_modes.py 256 2>&1 | grep -v Warning | tee $O/m256.txt
LANES=4,8 BIG=1 GPU_MAX_HW_QUEUES=8 timeout 300 python tools/lane_modes.py 256 2>&1 | grep -v Warning | tee $O/m256_q8.txt
LANES=2 BIG=0,1 timeout 300 python tools/lane_modes.py 512 2>&1 | grep -v Warning | tee $O/m512.txt
