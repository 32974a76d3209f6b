cd $GRAFT_REPO_ROOT
for v in "" 1 "" 1; do DFMIR_CONV3D_NO_WGRAD_MARCH=$v timeout 300 python scripts/bench_3d.py 2>&1 | grep -E 'ms/step' | cut -c1-90; done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -k "conv3d or registration3d" 2>&1 | tail -3
