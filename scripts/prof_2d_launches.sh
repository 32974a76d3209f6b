#!/bin/bash
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt2dl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-3d --roofline-steps 0 --host-input-steps 0 > $O/kt.log 2>&1
python $R/scripts/cs_launch_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) > $O/launches.txt 2>&1
rm -rf $O/kt
