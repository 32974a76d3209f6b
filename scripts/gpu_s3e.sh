#!/bin/bash
mkdir -p gpurun_out/s3e
(for v in "" _koepi ""; do DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip$v.so; [ -z "$v" ] && DFMIR_HIP_LIB=$PWD/dfmir_amd/libdfmir_hip.so; export DFMIR_HIP_LIB; python scripts/bench_wgrad2d.py fwd; done) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/s3e/fwd.txt
cat gpurun_out/s3e/fwd.txt
