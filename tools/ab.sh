#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
for rep in 1 2; do
run "default              " X=1
run "wgrad first          " BTC_WGRAD_FIRST=1
run "no deferred join     " BTC_DEFER_WGRAD=0
run "row order off        " BTC_ROW_ORDER=0
run "two-barrier wgrad    " BTC_TUNE=11=1
done
EXTRA="--features bf16"
for rep in 1 2; do
run "bf16 default         " X=1
run "bf16 dgrad first     " BTC_WGRAD_FIRST=0
done
