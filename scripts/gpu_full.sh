#!/bin/bash
# full GPU check of the tree: every gpu test, then the bench line
mkdir -p gpurun_out/full
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/full/tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
tail -5 gpurun_out/full/tests.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('2D: %.1f pairs/s %.2f ms  median %.2f [%.2f..%.2f]  fwd issued %.3f  wgrad issued %.3f' % (d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['step_ms']['max'], r['issued_frac'], r['wgrad_issued_frac']))
a=d['also_3d']; print('3D: %.2f ms; rough %s' % (a['ms_per_step'], a.get('rough_field')))
print('3D128: %.2f ms' % d['also_3d_128']['ms_per_step'])
print('warp:', d['roofline_hbm']['frac'], d['roofline_hbm']['bwd']['frac'], d['roofline_hbm']['rough_field']['fwd']['frac'], d['roofline_hbm']['rough_field']['bwd']['frac'])
print('cpu:', {k:(v['value'] if isinstance(v,dict) else v) for k,v in d['cpu_baseline'].items() if k in ('value','also_3d_128','also_3d_big')})
PY
