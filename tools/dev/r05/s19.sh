cd $GRAFT_REPO_ROOT
export YS_F8_MIN_CIN=32 YS_F8_MIN_TAPS=1 YS_GEMM_MIN_M=1 YS_HALO_MIN_FILL=1 YS_WGEMM_MIN_M=1
run() { python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['loss_items'])"; }
echo "== suite env"; for i in 1 2 3 4 5 6; do run; done
echo "== GEMM_HALO=0"; for i in 1 2 3 4; do YS_GEMM_HALO=0 run; done
echo "== BN_ATOMIC=0"; for i in 1 2 3 4; do YS_BN_ATOMIC=0 run; done
echo "== both off"; for i in 1 2 3 4; do YS_GEMM_HALO=0 YS_BN_ATOMIC=0 run; done
