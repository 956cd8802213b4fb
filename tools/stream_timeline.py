"""Per-stream busy / gap picture of the steady state from a rocprofv3 --kernel-trace CSV:
    python tools/stream_timeline.py <..._kernel_trace.csv> [lo=0.4] [hi=0.95]
per stream: launches and busy ms per step, the idle gaps between consecutive kernels of that stream by (previous kernel -> next kernel)
pair, and the union of busy intervals over all streams.  A step = 27 bn_apply launches of the forward pass."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = (float(sys.argv[2]) if len(sys.argv) > 2 else 0.4), (float(sys.argv[3]) if len(sys.argv) > 3 else 0.95)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
short = lambda n: re.sub(r"[<(].*", "", n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", ""))[:30]
K = [(short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id")), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
steps = max(sum(1 for k in K if k[0] == "bn_apply") / 27.0, 1.0)
t0, t1 = K[0][2], max(k[3] for k in K)
print("window %.1f ms, %.1f steps -> %.3f ms per step under the profiler" % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps))
by = collections.defaultdict(list)
for k in K:
    by[k[1]].append(k)
ev = sorted([(k[2], 1) for k in K] + [(k[3], -1) for k in K])
act, last, union = 0, None, 0
for t, d in ev:
    if act > 0:
        union += t - last
    act += d
    last = t
print("union of busy intervals %.3f ms/step, all streams idle %.3f ms/step" % (union / 1e6 / steps, ((t1 - t0) - union) / 1e6 / steps))
for s, ks in sorted(by.items(), key=lambda x: -sum(k[3] - k[2] for k in x[1])):
    busy = sum(k[3] - k[2] for k in ks)
    fam = collections.Counter()
    for k in ks:
        fam[k[0]] += k[3] - k[2]
    print("stream %s: %.0f launches/step, busy %.3f ms/step; top: %s" % (s, len(ks) / steps, busy / 1e6 / steps,
          ", ".join("%s %.3f" % (n, v / 1e6 / steps) for n, v in fam.most_common(8))))
    gaps = collections.defaultdict(lambda: [0, 0])
    tot = 0
    for a, b in zip(ks[:-1], ks[1:]):
        g = b[2] - a[3]
        if g > 0:
            gaps[(a[0], b[0])][0] += 1
            gaps[(a[0], b[0])][1] += g
            tot += g
    print("   gaps %.3f ms/step; largest by pair: %s" % (tot / 1e6 / steps, "; ".join("%s->%s %.3f ms (%.1f x, avg %.1f us)" % (k[0], k[1], v[1] / 1e6 / steps, v[0] / steps, v[1] / 1e3 / v[0])
                                                     for k, v in sorted(gaps.items(), key=lambda x: -x[1][1])[:10])))
