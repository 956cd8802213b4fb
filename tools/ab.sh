#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test
run() { # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
D="BTC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
for rep in 1 2; do
run "det fwd beside occ bwd       " BTC_SPLIT_BACKWARD=1
run "  + 8 hardware queues        " BTC_SPLIT_BACKWARD=1 GPU_MAX_HW_QUEUES=8
run "  + walk in front            " BTC_SPLIT_BACKWARD=1 BTC_DET_WALK_ASYNC=0
run "dist: split backward         " $D
run "dist: det fwd beside occ bwd " $D BTC_SPLIT_BACKWARD=1
done
EXTRA="--features bf16"
run "bf16 default                 " X=1
run "bf16 det fwd beside occ bwd  " BTC_SPLIT_BACKWARD=1
