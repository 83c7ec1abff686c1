# r06 session 2: batched weight-gradient hand-over (per-unit dy buffers, one event per batch, per-segment split reduction on the second stream): correctness, A/B sweep, trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s02; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_dist.py tests/test_model.py tests/test_bnred.py tests/test_abi.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -5 > $O/pytest_core.txt; cat $O/pytest_core.txt
S="--steps 40 --warmup 8 --no-cpu-baseline --no-nms --no-infer"
run() { tag=$1; shift; env "$@" python bench.py $S $LIBARG > $O/$tag.json 2> $O/$tag.err; python -c "import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['loss_items'])"; }
for rep in 1 2; do
LIBARG="--lib $R/build/libyolosharp_hip_r05.so" run prev_$rep A=1
LIBARG="" run b1_$rep YS_WG_BATCH=1
LIBARG="" run b6m40_$rep A=1
LIBARG="" run b4m24_$rep YS_WG_BATCH=4 YS_WG_BATCH_MB=24
LIBARG="" run b12m80_$rep YS_WG_BATCH=12 YS_WG_BATCH_MB=80
LIBARG="" run b100_$rep YS_WG_BATCH=100 YS_WG_BATCH_MB=100000
LIBARG="" run b8m200_$rep YS_WG_BATCH=8 YS_WG_BATCH_MB=200
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr1 -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-nms --no-infer > $O/tr1.log 2>&1
cd $R
python tools/dev/r06/stream_trace.py $O/tr1/t_kernel_trace.csv | tee $O/stream_trace.txt
timeout 1500 python -m pytest tests/test_production_routing.py -m gpu -q --no-header -p no:cacheprovider -k "other_configs" 2>&1 | tail -30 > $O/prod_routing_pytest.txt
tail -12 $O/prod_routing_pytest.txt
