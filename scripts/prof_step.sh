#!/bin/bash
# GPU box: kernel-by-kernel trace of the LAST step of a step script.  usage: prof_step.sh <script.py> <tag> [rows] [ADAMS]
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/$1 > $O/log.txt 2>&1
ADAMS=${4:-1} python $R/scripts/step_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) ${3:-60} > $O/step_trace.txt 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/kt
grep -E "ms/step|SUMMARY|us " $O/log.txt | tail -5; cat $O/step_trace.txt
