#!/bin/bash
# GPU box: HBM-side bytes of EVERY kernel of the 160x192x224 step (PMC passes over scripts/bench_3d.py)
export ONLY=160x192
bash scripts/prof_cmd.sh "python $GRAFT_REPO_ROOT/scripts/bench_3d.py" "" step3d > gpurun_out/pmc_step3d.txt 2>&1
rm -rf gpurun_out/prof_step3d/kt gpurun_out/prof_step3d/p1 gpurun_out/prof_step3d/p2 gpurun_out/prof_step3d/p3 gpurun_out/prof_step3d/p4
wc -l gpurun_out/pmc_step3d.txt
