"""GPU idle-gap analysis of a rocprofv3 kernel trace: where does the GPU wait for the host?"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last 60% of the trace (steady state)
rows = rows[int(len(rows) * 0.4):]
busy = 0; gaps = collections.defaultdict(lambda: [0, 0.0]); idle = 0
prev_end = None; prev_name = None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += en - st
    if prev_end is not None and st > prev_end:
        g = st - prev_end
        idle += g
        key = re.sub(r"<.*", "", prev_name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", ""))[:40] + " -> " + \
              re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", ""))[:40]
        gaps[key][0] += 1; gaps[key][1] += g
    prev_end = max(prev_end or 0, en); prev_name = r["Kernel_Name"]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span %.2f ms busy %.2f ms idle %.2f ms  (%d kernels)" % (span / 1e6, busy / 1e6, idle / 1e6, len(rows)))
hist = collections.Counter()
for k, (c, g) in gaps.items():
    pass
for k, (c, g) in sorted(gaps.items(), key=lambda x: -x[1][1])[:25]:
    print("%7.3f ms  n=%5d  avg %6.1f us  %s" % (g / 1e6, c, g / c / 1e3, k))
