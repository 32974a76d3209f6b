#!/bin/bash
# round 6: conv1x1_fwd_k -- tests + A/B
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d or random_geometries" > gpurun_out/r06/t_1x1fwd.txt 2>&1; tail -n 3 gpurun_out/r06/t_1x1fwd.txt
python -m pytest tests/test_gpu_models.py tests/test_gpu_nce_head.py -x -q -k "generator_golden or patch_sampler or whole_step_golden or fused_head or batch16_step" > gpurun_out/r06/t_1x1fwd_models.txt 2>&1; tail -n 3 gpurun_out/r06/t_1x1fwd_models.txt
for sw in NONE DFMIR_NO_1X1_FWD NONE DFMIR_NO_1X1_FWD; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; done > gpurun_out/r06/ab_1x1_fwd.txt 2>&1; cat gpurun_out/r06/ab_1x1_fwd.txt
