"""Aggregate the LAST train step of a rocprofv3 --kernel-trace CSV: busy fraction, time per kernel, torch elementwise
kernels bucketed by duration.  usage: python scripts/step_trace.py <kernel_trace.csv> [rows] ; ADAMS=<Adam launches per step, default 3>"""
import csv, os, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
NA = int(os.environ.get("ADAMS", 3))
ends = [i for k, i in enumerate(adam) if k % NA == NA - 1]     # a step ends with its Adam launches (3 in the 2-D step)
step = rows[ends[-2] + 1:ends[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print("last step: %d kernels, span %.2f ms, busy %.2f ms (%.1f%%)" % (len(step), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    nm = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = nm[:70]
    if "at::native" in nm:
        b = "<10us" if d < 10000 else ("<40us" if d < 40000 else ">=40us")
        what = "add" if "CUDAFunctor_add" in nm else "fill" if "Fill" in nm else nm[nm.find("at::native::") + 12:][:50]
        key = "torch " + what + " " + b
    agg[key][0] += 1; agg[key][1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    print("%7.3f ms %4d  %s" % (v[1] / 1e6, v[0], k))
