#!/bin/bash
# round 5: Waymo-shaped lines (fp32 / bf16) + KITTI bf16 / fp32 pairs, canary after the _borrow change
out=gpurun_out/r5i; mkdir -p $out
cd /root/repo
timeout 300 python -m pytest tests/test_hip_borrow_canary.py tests/test_hip_prefetch.py -q -m gpu -s 2>&1 | tail -3
for w in "--workload waymo" "--workload waymo --features bf16"; do
  timeout 400 python bench.py $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > $out/b.json
  python - "$w" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5i/b.json").read()); r = d.get("roofline") or {}; o = d.get("other_kernels") or {}
print(sys.argv[1], "%.1f scenes/s %.2f ms | conv_apply %.2f ms frac %.3f | wgrad %s | rulebook %s" % (d["value"], d["ms_per_step"], r.get("kernel_ms_per_step") or 0, r.get("frac") or 0,
      {k: round(v, 2) if isinstance(v, float) else v for k, v in (o.get("conv_wgrad") or {}).items() if k in ("GB/s", "ms_per_step")},
      {k: round(v, 2) if isinstance(v, float) else v for k, v in (o.get("rulebook") or {}).items() if k in ("GB/s", "ms_per_step")}))
PY
done
