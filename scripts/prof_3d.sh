#!/bin/bash
# GPU box: per-kernel times of the 3-D steps (scripts/bench_3d.py) -> gpurun_out/kt3d/stats.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt3d; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/scripts/bench_3d.py > $O/log.txt 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/stats.csv; rm -rf $O/kt
grep "ms/step" $O/log.txt
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats.csv")))
for r in sorted(rows,key=lambda r:-float(r["Percentage"]))[:${1:-22}]:
    print("%6d %10.1f us avg %7.2f%%  %s"%(int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["Percentage"]), r["Name"][:100]))
PY
