#!/bin/bash
O=gpurun_out/r06flow3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "resize or vecint or absmax or probe or split" > $O/pytest_ops.txt 2>&1; tail -n 3 $O/pytest_ops.txt
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -k "vxm or 3d or probe_audit" > $O/pytest_models3d.txt 2>&1; tail -n 3 $O/pytest_models3d.txt
for rep in 1 2; do
  for v in new base; do
    unset DFMIR_HIP_LIB DFMIR_CONV3D_NO_FLOW_MARCH
    [ $v = base ] && export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_r06base.so && export DFMIR_CONV3D_NO_FLOW_MARCH=1
    python scripts/bench_3d.py 2>/dev/null | cut -c1-64 | sed "s/^/$v  /" | tee -a $O/ab3d.txt
  done
done
unset DFMIR_HIP_LIB DFMIR_CONV3D_NO_FLOW_MARCH
bash scripts/prof_3d_step.sh 70 > $O/prof3d.txt 2>&1; cp gpurun_out/kt3d/step_trace.txt $O/step_trace_3d.txt; grep "last step\|resize\|absmax" $O/step_trace_3d.txt
