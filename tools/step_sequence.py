"""One step's kernels in launch order from a rocprofv3 kernel-trace CSV (in-order run: one stream): short name, duration -- which launch sits
between which.   python tools/step_sequence.py <kernel_trace.csv> [step_index_from_end=3]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
ends = [i for i, r in enumerate(rows) if "adam_apply" in r["Kernel_Name"]]
# a step ends with the second adam_apply (two parameter groups)
ends = ends[1::2]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
tot = 0.0
for r in rows[lo:hi]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n) if not n.startswith("elementwise") and "vectorized" not in n else n[:150]
    print("%8.1f +%6.1f gap %6.1f us  %s" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, n[:150]))
    prev_end = max(prev_end, b)
    tot += (b - a) / 1e3
print("launches %d, kernel us %.1f, span us %.1f" % (hi - lo, tot, (prev_end - t0) / 1e3))
