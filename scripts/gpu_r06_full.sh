#!/bin/bash
O=gpurun_out/r06full; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -n 5 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2 | tee $O/smoke.txt
python scripts/conv3d_step_census.py 2>&1 | grep -v amdgpu.ids > $O/conv3d_step_census.txt; cat $O/conv3d_step_census.txt
for v in new nos2; do
  unset DFMIR_CONV3D_NO_S2
  [ $v = nos2 ] && export DFMIR_CONV3D_NO_S2=1
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pil-workers 0 2>/dev/null | tail -n 1 > $O/line_$v.json
  python - $O/line_$v.json $v <<'PY' | tee -a $O/ab.txt
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[2], "2-D %.2f ms/step %.1f pairs/s | 3-D 160x192x224 %.3f ms | 128^3 %.3f ms" % (
    r["ms_per_step"], r["value"], r["also_3d"]["ms_per_step"], r["also_3d_128"]["ms_per_step"]))
PY
done
