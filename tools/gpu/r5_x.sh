cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q tests/test_frontend_gpu.py tests/test_embedding_gpu.py -k "full_batch" 2>&1 | tail -4
