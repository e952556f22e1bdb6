# round 5, call A: new tests (guard bands, pipelined fine-tune, surface, all-reduce ordering, streaming retry), fine-tune A/B, and the
# timing build of the 2-clip chain that died once in round 4 (profiles/r04_notes.md 1.4) -- three repetitions, rocgdb on a fault
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q -x tests/test_guard_bands_gpu.py tests/test_finetune_gpu.py tests/test_surface.py tests/test_streaming.py \
  "tests/test_train_gpu.py::test_allreduce_ranges_are_final_when_the_collective_reads_them" "tests/test_train_gpu.py::test_graph_replayed_training_step_equals_the_eager_step" \
  "tests/test_train_gpu.py::test_depthwise_batchnorm_as_one_operator" tests/test_pipeline_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for v in "1 0" "2 0" "2 1"; do
  set -- $v
  timeout 300 python bench.py --config finetune --no-cpu-baseline --ft-group $1 --ft-overlap $2 > $O/ft_g$1_o$2.json 2> $O/ft_g$1_o$2.err
  echo "finetune group=$1 overlap=$2 rc=$? $(python -c "
import json;d=json.load(open('$O/ft_g$1_o$2.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d['whole_step'])")"
done
timeout 300 python bench.py --config finetune --no-cpu-baseline --steps 20 --warmup 5 > $O/ft_driver.json 2> $O/ft_driver.err; echo "driver-style 20/5: $(python -c "
import json;d=json.load(open('$O/ft_driver.json'));print(d['value'],d['ms_per_step'])")"
# the fault of round 4: timing build, 512-clip handle (2-clip chain / 4-clip pair chain)
for rep in 1 2 3; do
  MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 200 python tools/chain_timing.py 512 > $O/timing_$rep.out 2> $O/timing_$rep.err
  rc=$?; echo "timing build, 512 clips, rep $rep: rc=$rc"
  if [ $rc -ne 0 ]; then
    tail -5 $O/timing_$rep.err
    MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 280 rocgdb -batch -ex "set pagination off" -ex run -ex "info threads" -ex bt \
      -ex "x/6i \$pc-12" -ex "info registers" --args python tools/chain_timing.py 512 > $O/rocgdb_$rep.log 2>&1
    grep -n "Thread.*stopped\|SIGSEGV\|SIGBUS\|aperture\|memory violation\|=>" $O/rocgdb_$rep.log | head -20
    break
  fi
done
