#!/bin/bash
mkdir -p gpurun_out/r06
{ python scripts/bench_conv1x1.py 2>&1 | grep -v amdgpu.ids; echo "--- DFMIR_NO_1X1_FWD=1 DFMIR_NO_1X1_WGRAD=1 (generic kernels)"; DFMIR_NO_1X1_FWD=1 DFMIR_NO_1X1_WGRAD=1 python scripts/bench_conv1x1.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/bench_conv1x1.txt; cat gpurun_out/r06/bench_conv1x1.txt
bash scripts/prof_cmd.sh "python $GRAFT_REPO_ROOT/scripts/bench_conv1x1.py" conv1x1 c1x1 > gpurun_out/r06/pmc_conv1x1.txt 2>&1; cat gpurun_out/r06/pmc_conv1x1.txt
