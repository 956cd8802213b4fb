cd /tmp && export TMPDIR=/tmp
for v in links off; do
  extra=""; [ $v = off ] && extra="BTC_TUNE=23=1"
  rm -rf /tmp/prof_$v
  env $extra BTC_SCHEDULE=in_order BTC_DEFER_WGRAD=0 BTC_OVERLAP_MIN_ROWS=2000000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o bench -- python /root/repo/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /tmp/prof_$v.json 2>/tmp/prof_$v.err
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep '^{' /tmp/prof_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = sum(int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]) / 2.0
tot = sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print("steps", steps, "kernel us/step %.1f launches/step %.1f" % (tot, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("conv_apply_s", "split_reduce", "bn_bwd_stats", "bn_bwd_apply")):
        print("  %-60s %6.2f calls/step %8.1f us/step avg %7.2f" % (n[26:86], int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, float(r["AverageNs"]) / 1e3))
PY
done
