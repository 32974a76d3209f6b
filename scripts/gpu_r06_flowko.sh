#!/bin/bash
# GPU box: knock-outs and per-phase trace of the march kernel's FLOW form (flow head forward at 160x192x224)
O=gpurun_out/r06flowko; rm -rf $O; mkdir -p $O
export ONLY_FLOW=1
for v in - m3ko1 m3ko2 m3ko4 m3ko6 m3ko8 m3koX32; do
  if [ "$v" = "-" ]; then unset DFMIR_HIP_LIB; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  python scripts/bench_flow_head.py 2>&1 | grep "march FLOW" | sed "s/^/$v  /" | tee -a $O/ko.txt
done
DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_m3trace.so python scripts/march_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
