# Round-3 GPU session 20: full GPU suite on the final kernels, then the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03t; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1; tail -6 $O/tests.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
