#!/bin/bash
# fused BatchNorm finalize + apply: quick safety run (short timeouts), then tests, then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/fin; mkdir -p $O
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-infer > $O/quick.json 2> $O/quick.err; echo "quick rc=$?"; tail -c 200 $O/quick.json; tail -3 $O/quick.err
timeout 900 python -m pytest tests/test_model.py tests/test_blocks.py tests/test_trainer.py -q -m gpu -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
bash tools/dev/r04_ab.sh fin - "fuse:-: sep:-:YS_BN_FUSE_FIN=0 fuse:-: sep:-:YS_BN_FUSE_FIN=0"
