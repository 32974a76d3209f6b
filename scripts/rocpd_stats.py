"""Dump the per-kernel summary (`top_kernels` view) of a rocprofv3 rocpd SQLite file as CSV."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, int(tot * 1000), int(avg * 1000), round(pct, 3)])
print("wrote", out, len(rows), "kernels")
