#!/bin/bash
# GPU box: kernel-by-kernel trace of the LAST 160x192x224 step of scripts/bench_3d.py -> gpurun_out/kt3d/step_trace.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt3d; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/scripts/bench_3d.py > $O/log.txt 2>&1
ADAMS=1 python $R/scripts/step_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) ${1:-60} > $O/step_trace.txt 2>&1
rm -rf $O/kt
grep "ms/step" $O/log.txt; cat $O/step_trace.txt
