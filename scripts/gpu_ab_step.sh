#!/bin/bash
# A/B of the whole 2-D step between the in-tree library and variant builds, same box, same call:
#   scripts/gpu_ab_step.sh <outdir> <tag> [<tag> ...]      ("-" = the in-tree library)
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then export DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; else export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-3d --roofline-steps 0 --host-input-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %.3f ms/step  median %.3f  min %.3f  %.1f pairs/s' % ('$v', d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['value']))" | tee -a $out/ab.txt
done; done
