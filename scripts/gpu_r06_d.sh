#!/bin/bash
# round 6, fourth box: the 1x1 weight-gradient kernel (tests + A/B) and the cost probe of a normalise-on-load in conv3x3_split_cs_k
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d" > gpurun_out/r06/t_conv2d.txt 2>&1; tail -3 gpurun_out/r06/t_conv2d.txt
python -m pytest tests/test_gpu_models.py tests/test_gpu_nce_head.py -x -q -k "generator_golden or patch_sampler or whole_step_golden or fused_head or deterministic" > gpurun_out/r06/t_1x1_models.txt 2>&1; tail -3 gpurun_out/r06/t_1x1_models.txt
for sw in NONE DFMIR_NO_1X1_WGRAD NONE DFMIR_NO_1X1_WGRAD; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; done > gpurun_out/r06/ab_1x1_wgrad.txt 2>&1; cat gpurun_out/r06/ab_1x1_wgrad.txt
P=$PWD/build/ko/libdfmir_hip_normprobe.so
{ for i in 1 2; do echo "--- product library"; python scripts/bench_conv.py 32 2>/dev/null | grep "^fwd"; echo "--- CS_NORM_PROBE (fma + max per staged value)"; DFMIR_HIP_LIB=$P python scripts/bench_conv.py 32 2>/dev/null | grep "^fwd"; done
  for lib in "" $P "" $P; do DFMIR_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-product}', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; done; } > gpurun_out/r06/cs_norm_probe.txt 2>&1; cat gpurun_out/r06/cs_norm_probe.txt
