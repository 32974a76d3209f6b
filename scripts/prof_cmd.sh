#!/bin/bash
# GPU box: kernel-trace + four PMC passes (separate runs, MI355X_MICROARCH.md) of an arbitrary command; prints the kernels whose
# name contains <substring>.   usage: scripts/prof_cmd.sh "<command>" <substring> <tag>   -> gpurun_out/prof_<tag>/
R=$GRAFT_REPO_ROOT; CMD="$1"; export PROF_SUB="$2"; export PROF_O=$R/gpurun_out/prof_$3; rm -rf $PROF_O; mkdir -p $PROF_O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $PROF_O/kt -- $CMD > $PROF_O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $PROF_O/p1 -- $CMD > $PROF_O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $PROF_O/p2 -- $CMD > $PROF_O/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $PROF_O/p3 -- $CMD > $PROF_O/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $PROF_O/p4 -- $CMD > $PROF_O/p4.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O, sub = os.environ["PROF_O"], os.environ["PROF_SUB"]
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            print(r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], r["AverageNs"])
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if sub in k:
                print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
