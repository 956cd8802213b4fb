#!/bin/bash
# back-to-back A/B bench lines in one box (value, p10 / median / p90 step ms); edit the run lines for the knobs under test.
# Knobs used for the numbers in DESIGN.md section 5: BTC_PIPELINE_OCC=0, BTC_SPLIT_BACKWARD=0, BTC_DET_WALK_ASYNC=0, BTC_FLAT_OPTIM=0,
# BTC_EARLY_OPT=1, BTC_ROW_ORDER=0|2, BTC_TUNE=<key>=<value>[,..] (tuning keys of include/btcdet_hip.h), BTC_PIN_CPUS=0, GPU_MAX_HW_QUEUES=8,
# and for the distributed path at world size 1:
#   D="BTC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0" [BTC_SYNC_BUCKETS=one|two] [BTC_SYNC_DRYRUN=1]
run() { # name, env assignments...
  name=$1; shift
  env "$@" python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['step_ms']['p10'], d['step_ms']['median'], d['step_ms']['p90'])"
}
for rep in 1 2 3; do
run "default                      " X=1
run "occupancy branch not ahead   " BTC_PIPELINE_OCC=0
run "one stream for both branches " BTC_SPLIT_BACKWARD=0
done
