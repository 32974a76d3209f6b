cd $GRAFT_REPO_ROOT
for v in "" 1 "" 1; do DFMIR_UPWGRAD_DIRECT=$v timeout 300 python scripts/bench_3d.py 2>&1 | grep -E 'ms/step' | cut -c1-90; done
