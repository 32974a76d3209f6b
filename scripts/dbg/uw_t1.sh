cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "upwgrad or upcat_conv3d_parity" 2>&1 | tail -3
for i in 1 2; do LEVELS=1 timeout 200 python scripts/bench_upwgrad.py 2>&1 | tail -1; done
timeout 600 bash scripts/prof_upconv3d.sh 2>&1 | grep "upwgrad4" | cut -c1-330
