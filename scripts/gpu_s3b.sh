#!/bin/bash
mkdir -p gpurun_out/s3b
(for d in 0 300 600 900 1200 1800 2400 0; do DFMIR_CS_DEPHASE=$d python scripts/bench_wgrad2d.py fwd; done) 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/s3b/fwd.txt
cat gpurun_out/s3b/fwd.txt
