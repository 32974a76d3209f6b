cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "ncc or losses" 2>&1 | tail -3
timeout 600 bash scripts/prof_3d_step.sh 70 2>&1 | grep -E "ncc|box_axis|ms/step" | cut -c1-100
