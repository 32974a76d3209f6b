#!/bin/bash
O=gpurun_out/r06flow2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "march or flow_head or resize" > $O/pytest_ops.txt 2>&1; tail -n 3 $O/pytest_ops.txt
ONLY_FLOW=1 python scripts/bench_flow_head.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_flow_head.txt
for rep in 1 2; do
  for v in new nofm; do
    unset DFMIR_CONV3D_NO_FLOW_MARCH
    [ $v = nofm ] && export DFMIR_CONV3D_NO_FLOW_MARCH=1
    python scripts/bench_3d.py 2>/dev/null | cut -c1-64 | sed "s/^/$v  /" | tee -a $O/ab3d.txt
  done
done
unset DFMIR_CONV3D_NO_FLOW_MARCH
bash scripts/prof_3d_step.sh 30 > $O/prof3d.txt 2>&1; cp gpurun_out/kt3d/step_trace.txt $O/step_trace_3d.txt; grep "march_k<16, 3\|last step\|resize" $O/step_trace_3d.txt
