cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -k "two_trainers or graph_replayed" 2>&1 | tail -3
