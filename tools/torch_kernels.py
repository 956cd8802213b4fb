"""torch's own kernels (at::native::*, rocclr copies / fills) in a rocprofv3 rocpd database, per stream: name (functor), launches per step,
us per step -- the launches between ours.   python tools/torch_kernels.py x_results.db [lo=0.3] [hi=0.9]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
lo, hi = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3, float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
rows = db.cursor().execute("select name, stream_id, queue_id, start, end from kernels order by start").fetchall()
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
steps = max(sum(1 for r in rows if "bn_stats" in r[0]) / 27.0, 1.0)
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, q, a, b in rows:
    if "at::native" in n or "rocclr" in n:
        m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda\w*|CatArray\w+|reduce_kernel|index\w+|rocclr_\w+|\w+Kernel\w*)", n.replace("at::native::", ""))
        key = (s, re.sub(r"<.*", "", n.replace("void ", "").replace("at::native::", ""))[:34] + " / " + (m.group(1)[:40] if m else "?"))
        agg[key][0] += 1
        agg[key][1] += b - a
for (s, k), (c, t) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print("stream %d  %-80s %5.1f launches/step %7.1f us/step" % (s, k, c / steps, t / 1e3 / steps))
