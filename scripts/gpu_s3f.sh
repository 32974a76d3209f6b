#!/bin/bash
mkdir -p gpurun_out/s3f
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv2d or split or reflect_conv" > gpurun_out/s3f/pytest.txt 2>&1
tail -2 gpurun_out/s3f/pytest.txt
(for v in "" _noearly _nt "" _noearly _nt; do DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip$v.so; [ -z "$v" ] && DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; export DFMIR_HIP_LIB; python scripts/bench_wgrad2d.py fwd; done) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/s3f/fwd.txt
cat gpurun_out/s3f/fwd.txt
