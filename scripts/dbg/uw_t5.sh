cd $GRAFT_REPO_ROOT
timeout 600 bash scripts/prof_3d_step.sh 70 2>&1 | grep -E "ncc|box_axis|ms/step" | cut -c1-100
