# does splitting the 1024-clip batch over concurrent graph branches (2 x 512, 4 x 256; same workgroup shapes) hide the kernels' ramps?
cd $GRAFT_REPO_ROOT
export PLANS=whole+pair
LANES=1 timeout 300 python tools/plan_sweep.py 1024 2>&1 | grep max_batch
LANES=1,2 BIG_TILES=1 timeout 300 python tools/plan_sweep.py 512 2>&1 | grep max_batch
LANES=2 BIG_TILES=0 timeout 300 python tools/plan_sweep.py 512 2>&1 | grep max_batch
LANES=4 BIG_TILES=1 timeout 300 python tools/plan_sweep.py 256 2>&1 | grep max_batch
