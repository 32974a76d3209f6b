python scripts/bench_flow_smooth.py 2>&1 | tail -n 1
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "smooth or grad_loss or flow or loss or edge" 2>&1 | tail -n 2
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "golden or 3d or fastcut" 2>&1 | tail -n 2
