# round 5, call B: the new tests again (guard bands fixed), fine-tune with the cheaper host draws, and the round-4 fault on the two
# revisions whose timing build spilled VGPRs in the 2-clip chain (_scratch/rev_*: git archive + make timing, not part of the tree)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_b; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest -m gpu -q tests/test_guard_bands_gpu.py tests/test_finetune_gpu.py tests/test_surface.py tests/test_streaming.py \
  "tests/test_train_gpu.py::test_allreduce_ranges_are_final_when_the_collective_reads_them" "tests/test_train_gpu.py::test_graph_replayed_training_step_equals_the_eager_step" \
  "tests/test_train_gpu.py::test_depthwise_batchnorm_as_one_operator" tests/test_pipeline_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
for v in "2 0" "1 0"; do
  set -- $v
  timeout 300 python bench.py --config finetune --no-cpu-baseline --ft-group $1 --ft-overlap $2 > $O/ft_g$1_o$2.json 2> $O/ft_g$1_o$2.err
  echo "finetune group=$1 overlap=$2 rc=$? $(python -c "
import json;d=json.load(open('$O/ft_g$1_o$2.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d['whole_step'])")"
done
timeout 300 python bench.py --config finetune --no-cpu-baseline --steps 20 --warmup 5 > $O/ft_driver.json 2> $O/ft_driver.err; echo "driver-style 20/5: $(python -c "
import json;d=json.load(open('$O/ft_driver.json'));print(d['value'],d['ms_per_step'])")"
timeout 300 python tools/finetune_group_profile.py > $O/ft_profile.txt 2>&1; cat $O/ft_profile.txt | head -30
for rev in 00dc737 593aa9c; do
  R=$GRAFT_REPO_ROOT/_scratch/rev_$rev
  [ -f $R/multilingual_kws_amd/lib/libmkws_hip_timing.so ] || { echo "no scratch build of $rev"; continue; }
  for rep in 1 2 3 4; do
    ( cd $R && MKWS_LIB=$R/multilingual_kws_amd/lib/libmkws_hip_timing.so PYTHONPATH=$R timeout 200 python tools/chain_timing.py 512 > $GRAFT_REPO_ROOT/$O/timing_${rev}_$rep.out 2> $GRAFT_REPO_ROOT/$O/timing_${rev}_$rep.err )
    rc=$?; echo "timing build of $rev, 512 clips, rep $rep: rc=$rc"
    if [ $rc -ne 0 ]; then
      grep -v "timing\]" $O/timing_${rev}_$rep.err | tail -6
      ( cd $R && MKWS_LIB=$R/multilingual_kws_amd/lib/libmkws_hip_timing.so PYTHONPATH=$R timeout 400 rocgdb -batch -ex "set pagination off" -ex run -ex "info threads" -ex bt \
        -ex "x/12i \$pc-24" -ex "info registers" --args python tools/chain_timing.py 512 > $GRAFT_REPO_ROOT/$O/rocgdb_${rev}.log 2>&1 )
      grep -n "stopped\|SIGSEGV\|SIGBUS\|SIGABRT\|aperture\|violation\|=>\|exec \|^#0\|^#1" $O/rocgdb_${rev}.log | head -30
      break
    fi
  done
done
