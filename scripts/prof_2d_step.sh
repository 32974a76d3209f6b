#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt2d; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-3d --roofline-steps 0 > $O/kt.log 2>&1
python $R/scripts/step_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) 70 > $O/step_trace.txt 2>&1
rm -rf $O/kt
cat $O/kt.log | tail -1 | cut -c1-300; cat $O/step_trace.txt
