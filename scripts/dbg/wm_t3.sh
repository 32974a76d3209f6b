cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad_march" 2>&1 | tail -2
for i in 1 2; do ONLY=32-16 timeout 300 python scripts/bench_conv3d.py 2>&1 | tail -1 | cut -c60-140; done
timeout 600 bash scripts/prof_conv3d.sh 32-16 2>&1 | grep "wgrad_march" | grep "p2\|p1" | cut -c1-400
