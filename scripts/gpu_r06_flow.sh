#!/bin/bash
# GPU box: the FLOW form of the march kernel + the 32-bit-index helpers: tests, micro-benchmarks, A/B of the 3-D steps
O=gpurun_out/r06flow; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "march or flow_head or upcat or vecint or resize or conv3d or warp" > $O/pytest_ops.txt 2>&1; tail -n 5 $O/pytest_ops.txt
python scripts/bench_flow_head.py 2>&1 | grep -v amdgpu.ids > $O/bench_flow_head.txt
DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_r06base.so python scripts/bench_flow_head.py 2>&1 | grep -v amdgpu.ids > $O/bench_flow_head_base.txt
cat $O/bench_flow_head.txt $O/bench_flow_head_base.txt
for rep in 1 2; do
  for v in new nofm base; do
    unset DFMIR_HIP_LIB DFMIR_CONV3D_NO_FLOW_MARCH
    [ $v = nofm ] && export DFMIR_CONV3D_NO_FLOW_MARCH=1
    [ $v = base ] && export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_r06base.so && export DFMIR_CONV3D_NO_FLOW_MARCH=1
    python scripts/bench_3d.py 2>/dev/null | cut -c1-64 | sed "s/^/$v  /" | tee -a $O/ab3d.txt
  done
done
unset DFMIR_HIP_LIB DFMIR_CONV3D_NO_FLOW_MARCH
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -k "vxm or 3d or probe_audit or skipping" > $O/pytest_models3d.txt 2>&1; tail -n 5 $O/pytest_models3d.txt
bash scripts/prof_3d_step.sh 70 > $O/prof3d.txt 2>&1; cp gpurun_out/kt3d/step_trace.txt $O/step_trace_3d.txt
