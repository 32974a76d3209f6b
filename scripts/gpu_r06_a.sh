#!/bin/bash
# round 6, first box: the new tests, the parity margins, a baseline bench line
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_ops.py -x -q -k "edge_branches or losses_golden or ncc" > gpurun_out/r06/t_edges.txt 2>&1; tail -3 gpurun_out/r06/t_edges.txt
python -m pytest tests/test_gpu_models.py -x -q -k "all_negatives" > gpurun_out/r06/t_allneg.txt 2>&1; tail -3 gpurun_out/r06/t_allneg.txt
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q -k "eight or captured_step" > gpurun_out/r06/t_dist8.txt 2>&1; tail -3 gpurun_out/r06/t_dist8.txt
DFMIR_MARGINS_OUT=gpurun_out/r06/margins_models.txt python -m pytest tests/test_gpu_models.py -q -s > gpurun_out/r06/t_models.txt 2>&1; tail -3 gpurun_out/r06/t_models.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06/bench0.json 2> gpurun_out/r06/bench0.err; python -c "
import json; d=json.loads(open('gpurun_out/r06/bench0.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['also_3d']['ms_per_step'], d['also_3d_128']['ms_per_step'], d['roofline_hbm'])"
