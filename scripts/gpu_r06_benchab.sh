#!/bin/bash
# GPU box: bench.py lines of the in-tree library against build/ko/libdfmir_hip_r06base.so (2-D step, 3-D steps under the graph)
O=gpurun_out/r06benchab; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  for v in new base; do
    unset DFMIR_HIP_LIB DFMIR_CONV3D_NO_FLOW_MARCH
    [ $v = base ] && export DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_r06base.so && export DFMIR_CONV3D_NO_FLOW_MARCH=1
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pil-workers 0 2>/dev/null | tail -n 1 > $O/line_$v$rep.json
    python - $O/line_$v$rep.json $v <<'PY' | tee -a $O/ab.txt
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[2], "2-D %.2f ms/step %.1f pairs/s | 3-D 160x192x224 %.3f ms | 128^3 %.3f ms" % (
    r["ms_per_step"], r["value"], r["also_3d"]["ms_per_step"], r["also_3d_128"]["ms_per_step"]))
PY
  done
done
