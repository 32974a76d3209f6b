#!/bin/bash
# round 6, third box: deterministic weight gradients + regression check of the default path
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_models.py -x -q -k "deterministic" > gpurun_out/r06/t_det.txt 2>&1; tail -5 gpurun_out/r06/t_det.txt
python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r06/t_ops.txt 2>&1; tail -3 gpurun_out/r06/t_ops.txt
python -m pytest tests/test_gpu_models.py -x -q -k "all_negatives or second_stream or registration3d_step or whole_step_golden or vxm_golden" > gpurun_out/r06/t_models2.txt 2>&1; tail -3 gpurun_out/r06/t_models2.txt
for sw in NONE DFMIR_DETERMINISTIC_WGRAD; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step  host enqueue', round(r['host_enqueue_ms_per_step'],2))"; done > gpurun_out/r06/ab_det.txt 2>&1; cat gpurun_out/r06/ab_det.txt
for sw in NONE DFMIR_DETERMINISTIC_WGRAD; do echo $sw; env $sw=1 python scripts/bench_3d.py 2>/dev/null | cut -c1-100; done > gpurun_out/r06/ab_det_3d.txt 2>&1; cat gpurun_out/r06/ab_det_3d.txt
