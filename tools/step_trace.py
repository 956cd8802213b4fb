"""every kernel of ONE step (the last full one) from a rocprofv3 --kernel-trace CSV of the in-order schedule: name with template
arguments, duration, grid -- the per-layer picture behind the per-family sums.  usage: python tools/step_trace.py trace.csv [filter]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
# step boundary: adam_apply launches (2 per step: occupancy group, detection group)
idx = [i for i, r in enumerate(rows) if "adam_apply" in r["Kernel_Name"]]
a, b = idx[-5] + 1, idx[-3] + 1
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:60] if m else n[:60]
tot = 0.0
for r in rows[a:b]:
    nm = short(r["Kernel_Name"])
    if flt and flt not in nm:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%-62s %8.1f us  grid %9s x %s" % (nm, d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "")))
print("total %.1f us over the listed launches" % tot)
