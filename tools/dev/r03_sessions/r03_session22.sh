# Round-3 GPU session 22: the per-layer determinism test -- does it see the bug on the build that had it (8ecb8ea), and pass on the tree?
cd $GRAFT_REPO_ROOT
export REPS=7
echo "== build 8ecb8ea (batched store loop)"; timeout 600 python tools/dev/determinism_layer.py build/libyolosharp_hip_buggy.so 2>&1 | grep -v "raw conv" | tail -8
echo "== tree"; timeout 600 python -m pytest tests/test_conv.py -q -m gpu -k reruns 2>&1 | tail -2
