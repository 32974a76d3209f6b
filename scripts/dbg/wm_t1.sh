cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad_march" 2>&1 | tail -15
ONLY=32-16 timeout 300 python scripts/bench_conv3d.py 2>&1 | tail -2
DFMIR_CONV3D_NO_WGRAD_MARCH=1 ONLY=32-16 timeout 300 python scripts/bench_conv3d.py 2>&1 | tail -1
