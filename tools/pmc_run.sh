cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_embedding_gpu.py -m gpu -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES -f csv -d gpurun_out/pmc2 -o p2 -- python tools/one_fwd.py > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -f csv -d gpurun_out/pmc3 -o p3 -- python tools/one_fwd.py > gpurun_out/pmc3.log 2>&1
ls gpurun_out/pmc2 gpurun_out/pmc3; tail -3 gpurun_out/pmc3.log
