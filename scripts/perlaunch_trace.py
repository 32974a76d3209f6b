import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    nm = r["Kernel_Name"]
    if not any(s in nm for s in ("conv3d_s2_fwd", "s2c2", "conv_mfma_k", "conv_wgrad_mfma_k")):
        continue
    key = (nm[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += d
for k, v in agg.items():
    print("%-62s grid %8s %4s %4s  x%3d  avg %7.1f us" % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0] / 1e3))
