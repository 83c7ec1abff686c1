# r06 session 5: full GPU suite on the current tree + A/B against the round-5 library + tail trace
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_s05; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
for rep in 1 2 3; do
python bench.py $S --lib build/libyolosharp_hip_r05.so 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'])"
python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['loss_items'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr1 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer > $GRAFT_REPO_ROOT/$O/tr1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/r06/stream_trace.py $O/tr1/t_kernel_trace.csv | head -14
