#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, prefix-cmd..., --
  name=$1; shift
  "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
nproc; lscpu | grep -E "NUMA node|Model name|Socket" | head -6
for rep in 1 2 3 4; do
run "plain      " env X=1
run "pin 0-15   " taskset -c 0-15
run "pin 0-3    " taskset -c 0-3
run "q8 pin 0-15" env GPU_MAX_HW_QUEUES=8 taskset -c 0-15
done
