cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "upwgrad or upcat_conv3d_parity" 2>&1 | tail -3
for i in 1 2 3; do LEVELS=1 timeout 200 python scripts/bench_upwgrad.py 2>&1 | tail -1; done
