cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_lanemodes3; rm -rf $O; mkdir -p $O
run() { echo "## $*" | tee -a $O/q.txt; env "$@" timeout 300 python tools/lane_modes.py 256 2>&1 | grep max_batch | tee -a $O/q.txt; }
run MODES=eager LANES=4 BIG=1
run MODES=graphs,eager LANES=4 BIG=1
run MODES=fork,graphs,eager LANES=4 BIG=1
run MODES=eager LANES=4 BIG=1 GPU_MAX_HW_QUEUES=8
run MODES=eager LANES=4 BIG=1 GPU_MAX_HW_QUEUES=2
run MODES=eager LANES=4 BIG=1 ROUNDS=1000
run MODES=eager,join LANES=2 BIG=1
