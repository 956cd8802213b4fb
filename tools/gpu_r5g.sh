#!/bin/bash
# round 5: the BatchNorm-statistics epilogue without its per-workgroup L2 write-back: fused == unfused tests, bench fp32 / bf16, kernel stats
out=gpurun_out/r5g; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_glue.py tests/test_hip_bf16.py tests/test_hip_bf16_mfma.py tests/test_hip_golden_full.py tests/test_hip_det_backbone.py tests/test_hip_pipeline.py tests/test_hip_prefetch.py tests/test_hip_borrow_canary.py -q -m gpu > $out/t.txt 2>&1; tail -3 $out/t.txt
AB_STEPS=80 bash tools/ab_env.sh 2 "fp32:" > $out/ab.txt 2>&1
for r in 1 2; do timeout 300 python bench.py --features bf16 --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 run %.1f scenes/s %.3f ms' % (d['value'], d['ms_per_step']))" >> $out/ab.txt; done
cat $out/ab.txt
export TMPDIR=/tmp
for f in f32 bf16; do
  flag=""; [ $f = bf16 ] && flag="--features bf16"
  (cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5g$f -o bench -- python /root/repo/bench.py $flag --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial_$f.json 2> /root/repo/$out/serial_$f.err)
  find /tmp/prof_r5g$f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_${f}_kernel_stats.csv
done
python - <<'PY'
import csv
for f in ("f32", "bf16"):
    rows = list(csv.DictReader(open("gpurun_out/r5g/serial_%s_kernel_stats.csv" % f)))
    steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
    print(f, "in order: launches/step %.1f" % (sum(int(r["Calls"]) for r in rows) / steps), "kernel ms/step %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
    fam = {}
    for r in rows:
        n = r["Name"]
        k = "conv_apply" if "conv_apply" in n else "split_reduce" if "split_reduce" in n else "conv_wgrad" if ("conv_wgrad" in n or "wgrad_reduce" in n) else "bn" if "bn_" in n else "rulebook" if ("rb_" in n or "order_local" in n) else "other"
        a = fam.setdefault(k, [0, 0]); a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    for k, (c, t) in sorted(fam.items(), key=lambda x: -x[1][1]):
        print("   %-14s %6.1f launches  %8.1f us / step" % (k, c / steps, t / steps / 1e3))
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:10]:
        print("      %6.2f /step %8.1f us/step  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, r["Name"][:90]))
PY
