cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d gpurun_out/pmc4 -o p4 -- python tools/one_fwd.py > gpurun_out/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM -f csv -d gpurun_out/pmc5 -o p5 -- python tools/one_fwd.py > gpurun_out/pmc5.log 2>&1
ls gpurun_out/pmc4 gpurun_out/pmc5
