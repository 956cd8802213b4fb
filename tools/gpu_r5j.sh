#!/bin/bash
# round 5: 1024-thread BatchNorm backward statistics, parallel last-arriver sums (sumsq2, occ_loss): tests, bn_bench both kernels, bench, kernel stats
out=gpurun_out/r5j; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_core.py tests/test_hip_glue.py tests/test_hip_bf16.py tests/test_hip_golden_full.py tests/test_hip_occupancy.py tests/test_hip_pipeline.py tests/test_hip_prefetch.py -q -m gpu -k "batchnorm or bn or glue or loss or golden or prefetch or pipeline or hot_path" > $out/t.txt 2>&1; tail -3 $out/t.txt
timeout 200 python tools/bn_bench.py > $out/bn_wide.txt 2>&1; tail -16 $out/bn_wide.txt
NARROW=1 timeout 200 python tools/bn_bench.py > $out/bn_narrow.txt 2>&1; tail -3 $out/bn_narrow.txt
BF=1 timeout 200 python tools/bn_bench.py > $out/bn_wide_bf16.txt 2>&1; tail -2 $out/bn_wide_bf16.txt
AB_STEPS=80 bash tools/ab_env.sh 2 "fp32:" > $out/ab.txt 2>&1
cat $out/ab.txt
export TMPDIR=/tmp
for f in f32; do
  (cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5j$f -o bench -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial_$f.json 2> /root/repo/$out/serial_$f.err)
  find /tmp/prof_r5j$f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_${f}_kernel_stats.csv
done
python - <<'PY'
import csv, json
for f in ("f32",):
    rows = list(csv.DictReader(open("gpurun_out/r5j/serial_%s_kernel_stats.csv" % f)))
    steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
    d = json.loads([l for l in open("gpurun_out/r5j/serial_%s.json" % f) if l.startswith("{")][-1])
    print(f, "in order (under rocprof): %.1f scenes/s; launches/step %.1f" % (d["value"], sum(int(r["Calls"]) for r in rows) / steps), "kernel ms/step %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
        if any(k in r["Name"] for k in ("bn_", "sumsq2", "occ_loss", "col_sum")):
            print("      %6.2f /step %8.1f us/step  avg %6.1f  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, int(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, r["Name"][:80]))
PY
