#!/bin/bash
# GPU box: kernel time + PMC passes of one conv launch shape -> gpurun_out/conv_prof/  (args: kind Cin Cout H n)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/conv_prof; rm -rf $O; mkdir -p $O
ARGS="${1:-fwd} ${2:-256} ${3:-256} ${4:-64} ${5:-32} 3"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/scripts/bench_one.py $ARGS > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python $R/scripts/bench_one.py $ARGS > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/p2 -- python $R/scripts/bench_one.py $ARGS > $O/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/p3 -- python $R/scripts/bench_one.py $ARGS > $O/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p4 -- python $R/scripts/bench_one.py $ARGS > $O/p4.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/conv_prof"
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "conv" in k and "at::" not in k:
                print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
