cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "upcat_conv3d_parity" 2>&1 | tail -5
timeout 500 bash scripts/gpu_upwgrad_ko.sh - 2 14 15 2>&1 | grep -v amdgpu.ids
