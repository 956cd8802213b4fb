"""per-dispatch durations of selected kernels from a rocprofv3 --kernel-trace CSV (one step's worth, in launch order)"""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2].split(",")
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pat)]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
t0 = None
for r in sel[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    print("%10.1f us  +%8.1f  %-40s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", "")))
