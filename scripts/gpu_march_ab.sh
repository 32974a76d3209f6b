#!/bin/bash
# GPU box: the z-marching 3-D conv kernel (csrc/conv3dm.hip) -- its tests, then the full-resolution layer shapes with the
# march kernel on and off (DFMIR_CONV3D_NO_MARCH=1) in the same process order, then the 3-D step both ways.
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "march or conv3d" 2>&1 | tail -15
for sw in 0 1; do
  echo "== DFMIR_CONV3D_NO_MARCH=$sw"
  if [ $sw = 1 ]; then export DFMIR_CONV3D_NO_MARCH=1; else unset DFMIR_CONV3D_NO_MARCH; fi
  ONLY=32-16,16-16,16-32 timeout 600 python scripts/bench_conv3d.py 2>&1 | tail -5
done
unset DFMIR_CONV3D_NO_MARCH
} > gpurun_out/march_ab.txt 2>&1
cat gpurun_out/march_ab.txt
