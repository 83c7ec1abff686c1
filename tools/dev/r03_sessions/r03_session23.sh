# Round-3 GPU session 23: asynchronous backward-segment ends in the data-parallel step (one-rank process group): parity tests, then step time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_dist.py tests/test_model.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for i in 1 2; do
MASTER_ADDR=127.0.0.1 MASTER_PORT=2952$i RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 400 python bench.py --force-dist --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer 2>$O/dist$i.err | tail -1 > $O/dist$i.json
python -c "import json; j=json.loads(open('$O/dist$i.json').read()); print('force-dist', j['ms_per_step'], j['loss_items'])"
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 > $O/local$i.json
python -c "import json; j=json.loads(open('$O/local$i.json').read()); print('local     ', j['ms_per_step'], j['loss_items'])"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | cut -c1-200
