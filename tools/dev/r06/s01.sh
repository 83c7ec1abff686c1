# r06 session 1: baseline on today's box, value of the second stream, two-stream kernel traces, calibration run of the new production-routing tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s01; mkdir -p $O
S="--no-cpu-baseline --no-nms --no-infer"
cd $R
python bench.py --steps 40 --warmup 8 $S > $O/c2_overlap1.json 2> $O/c2_overlap1.err
YS_OVERLAP=0 python bench.py --steps 40 --warmup 8 $S > $O/c2_overlap0.json 2> $O/c2_overlap0.err
python bench.py --steps 40 --warmup 8 $S > $O/c2_overlap1_b.json 2>/dev/null
for f in c2_overlap1 c2_overlap0 c2_overlap1_b; do python -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['loss_items'])"; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr1 -o t -- python $R/bench.py --steps 6 --warmup 3 $S > $O/tr1.log 2>&1
YS_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr0 -o t -- python $R/bench.py --steps 6 --warmup 3 $S > $O/tr0.log 2>&1
ls -la $O/tr1/* $O/tr0/* | head
cd $R
timeout 2400 python -m pytest tests/test_production_routing.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > $O/prod_routing_pytest.txt
tail -30 $O/prod_routing_pytest.txt
