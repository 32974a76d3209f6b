#!/bin/bash
# round 6, second box: the staged step (single-stream pieces) -- probe, tests, bench A/B
mkdir -p gpurun_out/r06
python scripts/graph_split_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/graph_split_probe.txt; cat gpurun_out/r06/graph_split_probe.txt
python -m pytest tests/test_gpu_models.py -x -q -k "second_stream or captured_step_matches or failed_capture or whole_step_golden or all_negatives or fastcut or key_feature" > gpurun_out/r06/t_staged.txt 2>&1; tail -3 gpurun_out/r06/t_staged.txt
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q -k "two_ranks or eight or captured_step or buckets" > gpurun_out/r06/t_dist8.txt 2>&1; tail -3 gpurun_out/r06/t_dist8.txt
for i in 1 2 3 4; do python -m pytest tests/test_gpu_models.py -q -s -k "trajectory" 2>&1 | grep "trajectory:"; done > gpurun_out/r06/trajectory_repeats.txt; cat gpurun_out/r06/trajectory_repeats.txt
for sw in NONE DFMIR_NO_STAGED NONE DFMIR_NO_STAGED; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step  host enqueue', round(r['host_enqueue_ms_per_step'],2), 'ms  pil', r['value_pil_loader'], r['step_submission'])"; done > gpurun_out/r06/ab_staged.txt 2>&1; cat gpurun_out/r06/ab_staged.txt
