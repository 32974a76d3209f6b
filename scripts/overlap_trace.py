"""Concurrency of the LAST train step of a rocprofv3 --kernel-trace CSV: span, sum of kernel durations, time with >= 1 /
>= 2 kernels running, and the same per queue / stream.  usage: python scripts/overlap_trace.py <kernel_trace.csv>"""
import csv, os, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if os.environ.get("ONLY_2D"):          # bench.py runs its 3-D legs after the 2-D one: cut at the first 3-D kernel
    cut = [i for i, r in enumerate(rows) if "conv3d" in r["Kernel_Name"] or "warp_win_fwd_k<3>" in r["Kernel_Name"]]
    rows = rows[:cut[0]] if cut else rows
adam = [i for i, r in enumerate(rows) if "adam_k" in r["Kernel_Name"]]
NA = int(os.environ.get("ADAMS", 3))
ends = [i for k, i in enumerate(adam) if k % NA == NA - 1]
step = rows[ends[-2] + 1:ends[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, cover = 0, t0, collections.Counter()
for t, d in ev:
    cover[min(depth, 3)] += t - last
    last = t; depth += d
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print("last step: %d kernels, span %.2f ms, sum of durations %.2f ms" % (len(step), (t1 - t0) / 1e6, tot / 1e6))
print("  idle %.2f ms | exactly 1 kernel %.2f ms | 2 kernels %.2f ms | >= 3 kernels %.2f ms" % tuple(cover[i] / 1e6 for i in range(4)))
for key in ("Queue_Id", "Stream_Id"):
    if key in step[0]:
        q = collections.defaultdict(lambda: [0, 0])
        for r in step:
            q[r[key]][0] += 1; q[r[key]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        print("  by %s: " % key + ", ".join("%s: %d kernels %.2f ms" % (k, v[0], v[1] / 1e6) for k, v in sorted(q.items())))
# the kernels that ran beside another one, by name
par = collections.defaultdict(lambda: [0, 0])
act = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    act = [a for a in act if a[1] > s]
    for a in act:
        ov = min(e, a[1]) - s
        if ov > 0:
            par[r["Kernel_Name"][:60]][0] += 1; par[r["Kernel_Name"][:60]][1] += ov
    act.append((s, e))
for k, v in sorted(par.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %7.3f ms overlapped with an earlier-started kernel, %3d x  %s" % (v[1] / 1e6, v[0], k))
