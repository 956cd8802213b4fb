#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
for rep in 1 2 3; do
run "pipelined occupancy branch   " X=1
run "det fwd beside occ bwd only  " BTC_PIPELINE_OCC=0
done
EXTRA="--features bf16"
for rep in 1 2; do
run "bf16 pipelined               " X=1
run "bf16 not pipelined           " BTC_PIPELINE_OCC=0
done
EXTRA="--workload waymo"
run "waymo pipelined              " X=1
run "waymo not pipelined          " BTC_PIPELINE_OCC=0
