cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-nms --no-infer 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['loss_items'])"; }
export YS_GEMM_HALO=0 YS_BN_ATOMIC=0
echo "== base(halo0 atom0) only GEMM_MIN_M=1"; for i in 1 2 3 4 5 6 7 8; do YS_GEMM_MIN_M=1 run; done
echo "== base only WGEMM_MIN_M=1"; for i in 1 2 3 4 5 6 7 8; do YS_WGEMM_MIN_M=1 run; done
echo "== base production"; for i in 1 2 3 4 5 6 7 8; do run; done
unset YS_GEMM_HALO
echo "== halo1 atom0 production"; for i in 1 2 3 4 5 6 7 8; do run; done
echo "== halo1 atom0 HALO_MIN_FILL=1"; for i in 1 2 3 4 5 6 7 8; do YS_HALO_MIN_FILL=1 run; done
unset YS_BN_ATOMIC
echo "== halo1 atom1 HALO_MIN_FILL=1"; for i in 1 2 3 4 5 6 7 8; do YS_HALO_MIN_FILL=1 run; done
