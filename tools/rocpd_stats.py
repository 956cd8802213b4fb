"""Per-kernel statistics from a rocprofv3 rocpd database (this rocprofv3 writes .db unless --output-format csv is given):
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows)
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
for r in rows:
    out.append((r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]))
if len(sys.argv) > 2:
    csv.writer(open(sys.argv[2], "w", newline="")).writerows(out)
for r in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-90s %7s %12s %10s %6s" % (str(r[0])[:90], r[1], r[2], r[3], r[4]))
