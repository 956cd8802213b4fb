#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
for rep in 1 2 3; do
run "one-launch BN " X=1
run "two-launch BN " BTC_TUNE=12=1
done
EXTRA="--features bf16"
for rep in 1 2; do
run "bf16 one-launch" X=1
run "bf16 two-launch" BTC_TUNE=12=1
done
