# the whole -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_tests; rm -rf $O; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest.log
