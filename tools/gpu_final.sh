#!/bin/bash
# the measurement set of a round: bench lines (default / bf16 / Waymo shape / Waymo bf16), kernel statistics of the default and the
# in-order schedule, the two PMC passes, the straggler estimate -> gpurun_out/<tag>/ (copy what is to be judged into profiles/)
tag=${1:-r05}
mkdir -p gpurun_out/$tag
cd /root/repo
timeout 400 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
timeout 300 python bench.py --features bf16 --no-cpu-baseline --no-extras > gpurun_out/$tag/bench_bf16.json 2> gpurun_out/$tag/bench_bf16.err
timeout 300 python bench.py --workload waymo --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/$tag/bench_waymo.json 2> gpurun_out/$tag/bench_waymo.err
timeout 300 python bench.py --workload waymo --features bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/$tag/bench_waymo_bf16.json 2> gpurun_out/$tag/bench_waymo_bf16.err
# same-box pairs, 80 steps: fp32 / bf16 on the default schedule and in order
for f in f32 bf16; do
  flag=""; [ $f = bf16 ] && flag="--features bf16"
  timeout 300 python bench.py $flag --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/$tag/bench80_$f.json 2> gpurun_out/$tag/bench80_$f.err
  BTC_SCHEDULE=in_order timeout 300 python bench.py $flag --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/$tag/bench80_inorder_$f.json 2> gpurun_out/$tag/bench80_inorder_$f.err
done
SKIP_BENCH=1 bash tools/gpu_prof.sh $tag > gpurun_out/$tag/prof.log 2>&1
# kernel statistics of the in-order schedule for the other three configurations (bf16, Waymo shape, Waymo bf16)
for cfg in "bf16|--features bf16 --steps 100 --warmup 10" "waymo|--workload waymo --steps 40 --warmup 5" "waymo_bf16|--workload waymo --features bf16 --steps 40 --warmup 5"; do
  name=${cfg%%|*}; extra=${cfg#*|}
  (cd /tmp && export TMPDIR=/tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$name -o bench -- python /root/repo/bench.py $extra --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/$tag/serial_${name}_prof.json 2> /root/repo/gpurun_out/$tag/serial_${name}_prof.err)
  find /tmp/prof_${tag}_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/serial_${name}_kernel_stats.csv
done
bash tools/gpu_pmc.sh $tag > gpurun_out/$tag/pmc.log 2>&1
bash tools/gpu_pmc.sh $tag "--features bf16" _bf16 >> gpurun_out/$tag/pmc.log 2>&1
bash tools/gpu_pmc.sh $tag "--workload waymo" _waymo >> gpurun_out/$tag/pmc.log 2>&1
bash tools/gpu_pmc.sh $tag "--workload waymo --features bf16" _waymo_bf16 >> gpurun_out/$tag/pmc.log 2>&1
timeout 300 python tools/straggler.py 64 gpurun_out/$tag/straggler.json > gpurun_out/$tag/straggler.log 2>&1
bash tools/gpu_full_heads_prof.sh > gpurun_out/$tag/full_heads.log 2>&1; cp gpurun_out/full_heads/tail_full.txt gpurun_out/$tag/full_heads_tail.txt 2>/dev/null
for f in bench bench_bf16 bench_waymo bench_waymo_bf16 bench80_f32 bench80_bf16 bench80_inorder_f32 bench80_inorder_bf16; do
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/$tag/$f.json") if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("$f", d["value"], d["ms_per_step"], "rulebook GB/s", d.get("rulebook_hbm_GBps"), "conv ms", r.get("kernel_ms_per_step"), "frac", r.get("frac"), r.get("bound"), (d.get("config") or {}).get("in_order_scenes_per_s"), (d.get("config") or {}).get("with_rpn_heads"))
except Exception as e:
    print("$f failed", e)
PY
done
grep -h "^{" gpurun_out/$tag/pmc.log | cut -c1-600
tail -3 gpurun_out/$tag/straggler.log
timeout 1100 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1; tail -3 gpurun_out/$tag/pytest_gpu.log
