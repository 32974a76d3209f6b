#!/bin/bash
# GPU box: kernel times + PMC passes of the parity-class kernels at the top-level layer shape -> gpurun_out/upconv_prof/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/upconv_prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_upconv3d.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $CMD > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p4 -- $CMD > $O/p4.log 2>&1
grep -h "parity-class\|rel-L2" $O/kt.log
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/upconv_prof"
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], r["AverageNs"])
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "conv3d" in k and "wsplit" not in k:
                print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
