#!/bin/bash
# round 6: the flow head's weight gradient on conv3d_flow_wgrad_k -- tests + A/B
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_ops.py -x -q -k "conv3d" > gpurun_out/r06/t_conv3d.txt 2>&1; tail -n 3 gpurun_out/r06/t_conv3d.txt
python -m pytest tests/test_gpu_models.py -x -q -k "registration3d_step or deterministic_weight_gradients_3d or vxm_golden" > gpurun_out/r06/t_3d_models.txt 2>&1; tail -n 3 gpurun_out/r06/t_3d_models.txt
for sw in 0 1 0 1; do if [ $sw = 1 ]; then export DFMIR_CONV3D_NO_FLOW_WGRAD=1; else unset DFMIR_CONV3D_NO_FLOW_WGRAD; fi; echo "DFMIR_CONV3D_NO_FLOW_WGRAD=$sw"; python scripts/bench_3d.py 2>/dev/null | cut -c1-100; done > gpurun_out/r06/ab_flow_wgrad.txt 2>&1; unset DFMIR_CONV3D_NO_FLOW_WGRAD; cat gpurun_out/r06/ab_flow_wgrad.txt
ONLY=16-3 python scripts/bench_conv3d.py 2>/dev/null | tail -n 4; DFMIR_CONV3D_NO_FLOW_WGRAD=1 ONLY=16-3 python scripts/bench_conv3d.py 2>/dev/null | tail -n 4
