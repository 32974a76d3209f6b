#!/bin/bash
# round-end checks on one box: the GPU suite twice (tolerance tests against split-K atomics: flakiness shows here), smoke, and
# the default bench command with its wall clock
mkdir -p gpurun_out/final
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_$i.txt 2>&1; tail -1 gpurun_out/final/pytest_$i.txt; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$SECONDS; python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "default bench: $((SECONDS - T0)) s wall"
python -c "
import json; d=json.loads(open('gpurun_out/final/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['issued_frac'], d['roofline']['single_stream'], d['also_3d']['ms_per_step'], d['also_3d_128']['ms_per_step'])"
