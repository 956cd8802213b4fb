#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
for rep in 1 2 3; do
run "walk beside conv1   " X=1
run "walk in front       " BTC_DET_WALK_ASYNC=0
done
EXTRA="--features bf16"
for rep in 1 2; do
run "bf16 walk beside    " X=1
run "bf16 walk in front  " BTC_DET_WALK_ASYNC=0
done
