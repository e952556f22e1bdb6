cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -m gpu -q tests/test_surface.py 2>&1 | tail -6
