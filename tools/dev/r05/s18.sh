cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_configs.py tests/test_dist.py -m gpu -q -k "c3_v8s or bench_force or c4_ or determin" 2>&1 | grep -E "passed|failed|FAILED" | head -5; done
