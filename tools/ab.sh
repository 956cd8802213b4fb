#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, env..., args
  name=$1; shift
  env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
for rep in 1 2 3; do
run "fp32 order=1" BTC_ROW_ORDER=1
run "fp32 order=0" BTC_ROW_ORDER=0
run "fp32 order=1 q8" BTC_ROW_ORDER=1 GPU_MAX_HW_QUEUES=8
done
EXTRA="--features bf16"
for rep in 1 2 3; do
run "bf16 order=1" BTC_ROW_ORDER=1
run "bf16 order=0" BTC_ROW_ORDER=0
run "bf16 order=1 q8" BTC_ROW_ORDER=1 GPU_MAX_HW_QUEUES=8
done
