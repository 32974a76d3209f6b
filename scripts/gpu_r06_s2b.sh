#!/bin/bash
O=gpurun_out/r06s2b; rm -rf $O; mkdir -p $O
for mp in 2 4 8 16; do DFMIR_S2W_MINP=$mp python scripts/bench_s2.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_s2.txt; done
DFMIR_CONV3D_NO_S2=1 python scripts/bench_s2.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_s2.txt
