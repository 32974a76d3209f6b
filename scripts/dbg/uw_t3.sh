cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -k "3d or upcat or voxelmorph or registration3d" 2>&1 | tail -5
for v in "" 1; do DFMIR_CONV3D_NO_UPWGRAD=$v timeout 300 python scripts/bench_3d.py 2>&1 | grep -E 'ms/step' ; done
for v in "" 1; do DFMIR_CONV3D_NO_UPWGRAD=$v timeout 300 python scripts/bench_3d.py 2>&1 | grep -E 'ms/step' ; done
