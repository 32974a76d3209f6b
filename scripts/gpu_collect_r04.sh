#!/bin/bash
export TAG=r04
bash scripts/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/collect/pytest_gpu.txt 2>&1
tail -3 gpurun_out/collect/pytest_gpu.txt
tail -c 600 gpurun_out/collect/bench.json
