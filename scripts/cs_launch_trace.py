"""Per-launch durations of the dominant 2-D kernels in the LAST step of a rocprofv3 kernel trace of bench.py, in launch order:
python scripts/cs_launch_trace.py <kernel_trace.csv> -- which of the 66 forward/dgrad and 35 weight-gradient launches are slow."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last adam_k triple's predecessor: take the last 499-kernel window ending at the last adam_k
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_k")]
end = idx[-1]
prev = [i for i in idx if i < end - 100]
beg = prev[-1] + 1 if prev else 0
sel = rows[beg:end + 1]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    nm = r["Kernel_Name"]
    if "conv3x3_split_cs_k" in nm or "wgrad_split2" in nm or "reflect_ring" in nm:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print("%9.1f us  %-28s grid %6s x %s x %s  %8.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, nm.split("(")[0][-28:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], d))
