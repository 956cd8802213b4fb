"""Steady-state kernel statistics from a rocprofv3 --kernel-trace CSV: only the LAST `frac` of the traced window is aggregated (the
head holds MIOpen's find runs, allocator growth and warm-up), per kernel name: launches and microseconds per step.
usage: python tools/trace_tail.py kernel_trace.csv n_steps_in_tail [frac=0.05] [top=45]
n_steps_in_tail: how many training steps the tail window covers (the caller knows: timed steps / step time)."""
import collections
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2])
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
top = int(sys.argv[4]) if len(sys.argv) > 4 else 45
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
if frac <= 0:      # window given in ms from the end
    cut = t1 - int(-frac * 1e6)
else:
    cut = t1 - int((t1 - t0) * frac)
tail = [r for r in rows if r[0] >= cut]
agg = collections.defaultdict(lambda: [0, 0])
for a, b, n in tail:
    n = re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
    agg[n][0] += 1
    agg[n][1] += b - a
tot = sum(v[1] for v in agg.values())
print("tail window %.1f ms, %d launches (%.0f / step), kernel time %.2f ms / step" % ((t1 - cut) / 1e6, len(tail), len(tail) / steps, tot / 1e6 / steps))
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print("%-72s %7.1f /step %9.1f us avg %8.3f ms/step" % (n, c / steps, t / c / 1e3, t / 1e6 / steps))
