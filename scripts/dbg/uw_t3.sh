cd $GRAFT_REPO_ROOT
for v in 400000 0 400000 0 100000; do DFMIR_UPWGRAD_MIN_VOX=$v timeout 300 python scripts/bench_3d.py 2>&1 | grep -E 'ms/step' | cut -c1-90; done
