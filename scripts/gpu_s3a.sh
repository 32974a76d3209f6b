#!/bin/bash
# session baseline: bench line + gpu tests
mkdir -p gpurun_out/s3a
python bench.py --steps 20 --warmup 5 > gpurun_out/s3a/bench.json 2> gpurun_out/s3a/bench.err
tail -c 1500 gpurun_out/s3a/bench.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s3a/pytest.txt 2>&1
tail -5 gpurun_out/s3a/pytest.txt
