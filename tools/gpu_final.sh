#!/bin/bash
# The measurement set of a round -> gpurun_out/<tag>/ (copy what is to be judged into profiles/ as <tag>_*):
#   bench lines            default (the driver's command) / bf16 / Waymo shape / Waymo bf16; 80-step same-box repetitions, default + in order
#   kernel statistics      alone (in order, weight gradients in stream: nothing beside anything), serial (in order, weight gradients on
#                          their side stream), default schedule (+ stream timeline) -- alone + serial for all four configurations
#   one step's kernels     in launch order (alone)
#   PMC                    FETCH_SIZE / WRITE_SIZE passes x 4 configurations
#   straggler estimate, full-heads tail, the whole -m gpu suite
# usage: bash tools/gpu_final.sh [tag] [sections]     sections: any of bench,pairs,stats,pmc,misc,tests (default: all)
tag=${1:-r06}
sections=${2:-bench,pairs,stats,pmc,misc,tests}
out=gpurun_out/$tag
mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
has() { [[ ",$sections," == *",$1,"* ]]; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r, c = d.get("roofline") or {}, d.get("config") or {}
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "| rulebook GB/s", d.get("rulebook_hbm_GBps"), "| conv ms", r.get("kernel_ms_per_step"), "frac", r.get("frac"),
          "sched", r.get("frac_scheduled"), r.get("bound"), "| in order", c.get("in_order_scenes_per_s"), "| rpn", (c.get("with_rpn_heads") or {}).get("scenes_per_s"),
          "| all heads", (c.get("with_all_heads") or {}).get("scenes_per_s"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
if has bench; then
  timeout 500 python bench.py > $out/bench.json 2> $out/bench.err; line $out/bench.json
  timeout 300 python bench.py --features bf16 --no-cpu-baseline --no-extras > $out/bench_bf16.json 2> $out/bench_bf16.err; line $out/bench_bf16.json
  timeout 300 python bench.py --workload waymo --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_waymo.json 2> $out/bench_waymo.err; line $out/bench_waymo.json
  timeout 300 python bench.py --workload waymo --features bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_waymo_bf16.json 2> $out/bench_waymo_bf16.err; line $out/bench_waymo_bf16.json
fi
if has pairs; then   # same-box repetitions, 80 steps: the default schedule three times per precision (a process is fast or slow as a whole), in order once
  for f in f32 bf16; do
    flag=""; [ $f = bf16 ] && flag="--features bf16"
    for rep in a b c; do
      timeout 300 python bench.py $flag --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $out/bench80_${f}_$rep.json 2> $out/bench80_$f.err; line $out/bench80_${f}_$rep.json
    done
    BTC_SCHEDULE=in_order timeout 300 python bench.py $flag --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $out/bench80_inorder_$f.json 2> $out/bench80_inorder_$f.err; line $out/bench80_inorder_$f.json
  done
fi
stats() {   # stats <name> <env...> -- <bench args...>: rocprofv3 kernel statistics (+ trace) of one bench command
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  (cd /tmp && env "${envs[@]}" timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$name -o bench -- python /root/repo/bench.py "$@" --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/${name}_prof.json 2> /root/repo/$out/${name}_prof.err)
  find /tmp/prof_${tag}_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${name}_kernel_stats.csv
  python - $out/${name}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = sum(int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]) / 2.0
fam = {}
for r in rows:
    n = r["Name"]
    k = "conv_apply" if ("conv_apply" in n or "split_reduce" in n) else "conv_wgrad" if ("conv_wgrad" in n or "wgrad_reduce" in n) else "bn" if "bn_" in n else \
        "rulebook" if ("rb_" in n or "order_local" in n) else "other"
    a = fam.setdefault(k, [0, 0]); a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
print(sys.argv[1].split("/")[-1], "steps %d launches/step %.1f kernel ms/step %.3f |" % (steps, sum(int(r["Calls"]) for r in rows) / steps, sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6),
      " ".join("%s %.0f us / %.1f" % (k, t / steps / 1e3, c / steps) for k, (c, t) in sorted(fam.items(), key=lambda x: -x[1][1])))
PY
}
if has stats; then
  for cfg in "f32|--steps 100 --warmup 10" "bf16|--features bf16 --steps 100 --warmup 10" "waymo|--workload waymo --steps 40 --warmup 5" "waymo_bf16|--workload waymo --features bf16 --steps 40 --warmup 5"; do
    name=${cfg%%|*}; extra=${cfg#*|}; sfx="_$name"; [ $name = f32 ] && sfx=""
    stats alone$sfx BTC_SCHEDULE=in_order BTC_DEFER_WGRAD=0 BTC_OVERLAP_MIN_ROWS=2000000000 -- $extra
    stats serial$sfx BTC_SCHEDULE=in_order -- $extra
  done
  find /tmp/prof_${tag}_alone -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/step_sequence.py {} 3 > $out/alone_step_sequence.txt 2>&1; tail -1 $out/alone_step_sequence.txt
  find /tmp/prof_${tag}_serial -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/step_trace.py {} conv > $out/serial_step_conv.txt 2>&1
  stats bench A=1 -- --steps 40 --warmup 10
  find /tmp/prof_${tag}_bench -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/stream_timeline.py {} > $out/bench_timeline.txt 2>&1; head -c 400 $out/bench_timeline.txt; echo
  stats bench_bf16 A=1 -- --features bf16 --steps 40 --warmup 10
fi
if has pmc; then
  bash tools/gpu_pmc.sh $tag > $out/pmc.log 2>&1
  bash tools/gpu_pmc.sh $tag "--features bf16" _bf16 >> $out/pmc.log 2>&1
  bash tools/gpu_pmc.sh $tag "--workload waymo" _waymo >> $out/pmc.log 2>&1
  bash tools/gpu_pmc.sh $tag "--workload waymo --features bf16" _waymo_bf16 >> $out/pmc.log 2>&1
  grep -h "^{" $out/pmc.log | cut -c1-500
fi
if has misc; then
  timeout 300 python tools/straggler.py 64 $out/straggler.json > $out/straggler.log 2>&1; tail -2 $out/straggler.log
  bash tools/gpu_full_heads_prof.sh > $out/full_heads.log 2>&1; cp gpurun_out/full_heads/tail_full.txt $out/full_heads_tail.txt 2>/dev/null
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
  BTC_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_dist1.json 2> $out/bench_dist1.err; line $out/bench_dist1.json
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
fi
