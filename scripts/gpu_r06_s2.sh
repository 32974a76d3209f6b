#!/bin/bash
O=gpurun_out/r06s2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "stride2 or conv3d or tiny_volume" > $O/pytest_ops.txt 2>&1; tail -n 15 $O/pytest_ops.txt
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -k "vxm or 3d or probe_audit or skipping" > $O/pytest_models3d.txt 2>&1; tail -n 5 $O/pytest_models3d.txt
for rep in 1 2; do
  for v in new nos2; do
    unset DFMIR_CONV3D_NO_S2
    [ $v = nos2 ] && export DFMIR_CONV3D_NO_S2=1
    python scripts/bench_3d.py 2>/dev/null | cut -c1-64 | sed "s/^/$v  /" | tee -a $O/ab3d.txt
  done
done
unset DFMIR_CONV3D_NO_S2
bash scripts/prof_3d_step.sh 70 > $O/prof3d.txt 2>&1; cp gpurun_out/kt3d/step_trace.txt $O/step_trace_3d.txt; grep "last step\|s2_\|conv_mfma\|conv_wgrad_mfma\|absmax" $O/step_trace_3d.txt
