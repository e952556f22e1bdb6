cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/pmc_fetch -o f -- python tools/one_fwd.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d gpurun_out/pmc_write -o w -- python tools/one_fwd.py > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/pmc_fetch gpurun_out/pmc_write
