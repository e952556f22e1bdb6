cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence
mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_write -o w -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o m -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_mfma.log 2>&1 )
find $O -name "*counter_collection.csv" | head; tail -3 $O/pmc_mfma.log
