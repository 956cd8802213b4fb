"""Per-stream busy time, union of busy intervals and the largest idle gaps of the main stream from a rocprofv3 rocpd database
(steady-state window: fractions of the trace given on the command line):
    python tools/stream_gaps.py gpurun_out/prof/x_results.db [lo=0.3] [hi=0.9] [ms_per_step=6.5]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
lo, hi = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3, float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
rows = db.cursor().execute("select name, stream_id, queue_id, start, end from kernels order by start").fetchall()
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
t0, t1 = rows[0][3], max(r[4] for r in rows)
n_adam = sum(1 for r in rows if "FusedAdam" in r[0] or "fused_adam" in r[0].lower())
by, cnt = collections.defaultdict(float), collections.Counter()
for n, s, q, a, b in rows:
    by[(s, q)] += b - a
    cnt[(s, q)] += 1
main = max(by, key=lambda k: cnt[k])
steps = max(sum(1 for r in rows if "bn_stats" in r[0]) / 27.0, 1.0)
print("window %.1f ms, ~%.1f steps (27 bn_stats per step) -> %.2f ms per step under the profiler" % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps))
for k, v in sorted(by.items(), key=lambda x: -x[1]):
    print("  stream %s: busy %.2f ms/step, %.0f launches/step%s" % (k, v / 1e6 / steps, cnt[k] / steps, "  <- main" if k == main else ""))
fam = collections.defaultdict(float)
famc = collections.Counter()
short0 = lambda n: re.sub(r"[<(].*", "", n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", ""))[:40]
for n, s, q, a, b in rows:
    fam[((s, q), short0(n))] += b - a
    famc[((s, q), short0(n))] += 1
for k in sorted(by, key=lambda x: -by[x])[:3]:
    top = sorted(((v, n) for (kk, n), v in fam.items() if kk == k), reverse=True)[:12]
    print("  stream %s top kernels (ms/step, launches/step): " % (k,) + ", ".join("%s %.3f/%.0f" % (n, v / 1e6 / steps, famc[(k, n)] / steps) for v, n in top))
ev = sorted([(a, 1) for _, _, _, a, b in rows] + [(b, -1) for _, _, _, a, b in rows])
act, last, union = 0, None, 0
for t, d in ev:
    if act > 0:
        union += t - last
    act += d
    last = t
print("  union of busy intervals %.2f ms/step, all streams idle %.2f ms/step" % (union / 1e6 / steps, ((t1 - t0) - union) / 1e6 / steps))
mr = [r for r in rows if (r[1], r[2]) == main]
short = lambda n: re.sub(r"<.*", "", n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", ""))[:34]
gaps, tot = collections.defaultdict(lambda: [0, 0.0]), 0
for p, c in zip(mr[:-1], mr[1:]):
    g = c[3] - p[4]
    if g > 0:
        tot += g
        k = short(p[0]) + " -> " + short(c[0])
        gaps[k][0] += 1
        gaps[k][1] += g
print("  main-stream gaps %.2f ms/step; largest:" % (tot / 1e6 / steps))
for k, (c, g) in sorted(gaps.items(), key=lambda x: -x[1][1])[:14]:
    print("   %6.3f ms/step  %4.1f per step  avg %6.1f us  %s" % (g / 1e6 / steps, c / steps, g / c / 1e3, k))
