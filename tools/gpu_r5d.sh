#!/bin/bash
# round 5, fourth GPU session: the whole GPU suite on the cleaned-up build, bench lines (fp32 / bf16), kernel statistics (launches per step)
out=gpurun_out/r5d; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests -q -m gpu > $out/t_all.txt 2>&1; tail -12 $out/t_all.txt
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err
timeout 300 python bench.py --features bf16 --no-cpu-baseline --no-extras > $out/bench_bf16.json 2> $out/bench_bf16.err
SKIP_BENCH=1 bash tools/gpu_prof.sh r5d > $out/prof.log 2>&1; tail -5 $out/prof.log
python - <<'PY'
import json, csv
for f in ("bench", "bench_bf16"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5d/%s.json" % f) if l.startswith("{")][-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], "in-order", (d.get("config") or {}).get("in_order_scenes_per_s"), "conv ms", r.get("kernel_ms_per_step"), "frac", r.get("frac"),
              "wgrad", (d.get("other_kernels") or {}).get("conv_wgrad"), "rb", (d.get("other_kernels") or {}).get("rulebook"))
    except Exception as e:
        print(f, "failed", e)
try:
    rows = list(csv.DictReader(open("gpurun_out/r5d/serial_kernel_stats.csv")))
    steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
    tot = sum(int(r["Calls"]) for r in rows)
    print("serial: %.1f launches / step over %d steps, %.3f ms of kernel time / step" % (tot / steps, steps, sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
    for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:45]:
        print("%7.2f %8.1f us  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, r["Name"][:110]))
except Exception as e:
    print("stats failed", e)
PY
