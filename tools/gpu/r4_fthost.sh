cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_fthost
timeout 300 python tools/finetune_host_profile.py 2>&1 | cut -c1-200 | tee gpurun_out/r4_fthost/out.txt | head -60
