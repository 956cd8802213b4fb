#!/bin/bash
# kernel statistics of the step with NOTHING concurrent (in order, weight gradients in stream, no side stream): true kernel durations,
# the command of round 4's serial profile (profiles/r04_serial_kernel_stats.csv) -> gpurun_out/r5n/
out=gpurun_out/r5n; mkdir -p $out
export TMPDIR=/tmp
for f in f32 bf16; do
  flag=""; [ $f = bf16 ] && flag="--features bf16"
  (cd /tmp && BTC_SCHEDULE=in_order BTC_DEFER_WGRAD=0 BTC_OVERLAP_MIN_ROWS=2000000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5n$f -o bench -- python /root/repo/bench.py $flag --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/alone_$f.json 2> /root/repo/$out/alone_$f.err)
  find /tmp/prof_r5n$f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/$out/alone_${f}_kernel_stats.csv
done
cd /root/repo
python - <<'PY'
import csv, json
for f in ("f32", "bf16"):
    rows = list(csv.DictReader(open("gpurun_out/r5n/alone_%s_kernel_stats.csv" % f)))
    steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
    d = json.loads([l for l in open("gpurun_out/r5n/alone_%s.json" % f) if l.startswith("{")][-1])
    print(f, "%.1f scenes/s under rocprof; launches/step %.1f" % (d["value"], sum(int(r["Calls"]) for r in rows) / steps), "kernel ms/step %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
PY
