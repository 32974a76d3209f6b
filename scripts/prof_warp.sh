#!/bin/bash
# GPU box: kernel durations + three PMC passes for the warp kernels -> gpurun_out/warp_prof/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/warp_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/scripts/prof_warp.py > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python $R/scripts/prof_warp.py > $O/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/p2 -- python $R/scripts/prof_warp.py > $O/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p3 -- python $R/scripts/prof_warp.py > $O/p3.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/warp_prof"
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
for p in ("p1", "p2", "p3"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "warp" in k:
                print(p, k, {c: sum(v) / len(v) for c, v in d.items()})
PY
