#!/bin/bash
# round 5: bf16 against fp32 on one box (bench lines, launches per step), where the bf16 step's extra torch ops come from; canary
out=gpurun_out/r5f; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests -q -m gpu > $out/t.txt 2>&1; tail -4 $out/t.txt; grep "poisoned" $out/t.txt
AB_STEPS=80 bash tools/ab_env.sh 2 "fp32:" > $out/ab.txt 2>&1
for r in 1 2; do timeout 300 python bench.py --features bf16 --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 run %.1f scenes/s %.3f ms' % (d['value'], d['ms_per_step']))" >> $out/ab.txt; done
cat $out/ab.txt
export TMPDIR=/tmp
(cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5f -o bench -- python /root/repo/bench.py --features bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial_bf16.json 2> /root/repo/$out/serial_bf16.err)
find /tmp/prof_r5f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_bf16_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r5f/serial_bf16_kernel_stats.csv")))
steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
print("bf16 in order: steps", steps, "launches/step %.1f" % (sum(int(r["Calls"]) for r in rows) / steps), "kernel ms/step %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:40]:
    print("%6.2f /step %8.1f us/step  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, r["Name"][:100]))
PY
FEATURES=bf16 ALL_OPS=1 TOP=60 timeout 300 python tools/op_sites.py 2>&1 | grep -v "empty \|detach\|select \|slice \|view \| empty_like" | head -70
(cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5f32 -o bench -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial_f32.json 2> /root/repo/$out/serial_f32.err)
find /tmp/prof_r5f32 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_f32_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r5f/serial_f32_kernel_stats.csv")))
steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
print("fp32 in order: steps", steps, "launches/step %.1f" % (sum(int(r["Calls"]) for r in rows) / steps), "kernel ms/step %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:24]:
    print("%6.2f /step %8.1f us/step  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, r["Name"][:100]))
PY
