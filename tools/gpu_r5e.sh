#!/bin/bash
# round 5: rulebook chain timing after the adaptive-span bitmap marks; the _borrow canary; two-launch scans
out=gpurun_out/r5e; mkdir -p $out
cd /root/repo
timeout 600 python -m pytest tests/test_hip_chain_rulebooks.py tests/test_hip_core.py tests/test_hip_borrow_canary.py tests/test_hip_golden_full.py -q -m gpu -s > $out/t.txt 2>&1; tail -6 $out/t.txt; grep "poisoned" $out/t.txt
export TMPDIR=/tmp
(cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5e -o bench -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial.json 2> /root/repo/$out/serial.err)
find /tmp/prof_r5e -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r5e/serial_kernel_stats.csv")))
steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
print("steps", steps, "launches/step %.1f" % (sum(int(r["Calls"]) for r in rows) / steps), "kernel ms/step %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
for r in rows:
    if any(k in r["Name"] for k in ("rb_", "scan_", "order_local", "pov_", "vox_")):
        print("%6.2f /step %8.1f us/step  avg %7.1f  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, float(r["AverageNs"]) / 1e3, r["Name"][:80]))
PY
