cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_frontend_gpu.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --config frontend --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('frontend', d['value'], d['ms_per_step'], d['roofline']['frac'])"
