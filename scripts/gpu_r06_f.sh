#!/bin/bash
mkdir -p gpurun_out/r06
python scripts/bench_in_fuse_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/in_fuse_probe.txt; cat gpurun_out/r06/in_fuse_probe.txt
P=$PWD/build/ko/libdfmir_hip_normprobe2.so
{ for i in 1 2; do echo "--- product library"; python scripts/bench_conv.py 32 2>/dev/null | grep "^wgrad"; echo "--- W2_NORM_PROBE ((x - m) * r, max per converted value)"; DFMIR_HIP_LIB=$P python scripts/bench_conv.py 32 2>/dev/null | grep "^wgrad"; done
  for lib in "" $P "" $P; do DFMIR_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-product}', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; done; } > gpurun_out/r06/w2_norm_probe.txt 2>&1; cat gpurun_out/r06/w2_norm_probe.txt
