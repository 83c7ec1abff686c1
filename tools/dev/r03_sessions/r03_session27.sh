# Round-3 GPU session 27: does a forced delay between the K loop's last MFMA and the epilogue's first conversion change the determinism defect of the batched build (8ecb8ea)?
cd $GRAFT_REPO_ROOT; export REPS=12
for v in hz_plain hz_delay; do echo "== $v"; timeout 600 python tools/dev/determinism_layer.py build/libyolosharp_hip_$v.so 2>&1 | grep -E "differing|channels" | cut -c1-200; done
