cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_s13
timeout 1800 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r06_s13/pytest_gpu.txt; tail -5 gpurun_out/r06_s13/pytest_gpu.txt
bash tools/r06_all.sh
