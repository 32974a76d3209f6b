cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/uwprof -- python $R/scripts/bench_upwgrad.py > /tmp/uw.log 2>&1
tail -2 /tmp/uw.log
python - <<'PY'
import glob
for f in glob.glob("/tmp/uwprof/**/*kernel_stats.csv", recursive=True):
    for i, l in enumerate(open(f)):
        if i < 12: print(l.strip()[:160])
PY
