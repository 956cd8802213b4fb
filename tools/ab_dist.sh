#!/bin/bash
# A/B of the data-parallel path at world size 1 over RCCL (one bench line per variant, value + ms per step)
mkdir -p gpurun_out/ab
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab/$name.json"))
    print("%-28s %7.1f scenes/s  %6.3f ms/step  median %.3f  allocs %d" % ("$name", d["value"], d["ms_per_step"], d["step_ms"]["median"], d["step_ms"]["device_allocs_in_timed_region"]))
except Exception as e:
    print("$name failed", e)
PY
}
for v in "$@"; do
  case $v in
    nogroup) timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ab/nogroup.json 2> gpurun_out/ab/nogroup.err; python -c "import json; d=json.load(open('gpurun_out/ab/nogroup.json')); print('%-28s %7.1f scenes/s  %6.3f ms/step' % ('nogroup', d['value'], d['ms_per_step']))";;
    group_nosync_q8) run $v BTC_BENCH_FORCE_DIST=1 BTC_BENCH_NOSYNC=1;;
    group_nosync_q4) run $v BTC_BENCH_FORCE_DIST=1 BTC_BENCH_NOSYNC=1 GPU_MAX_HW_QUEUES=4;;
    direct_q8) run $v BTC_BENCH_FORCE_DIST=1;;
    direct_q4) run $v BTC_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4;;
    direct_dry_q8) run $v BTC_BENCH_FORCE_DIST=1 BTC_SYNC_DRYRUN=1;;
    direct_dry_q4) run $v BTC_BENCH_FORCE_DIST=1 BTC_SYNC_DRYRUN=1 GPU_MAX_HW_QUEUES=4;;
    direct_nocheck_q4) run $v BTC_BENCH_FORCE_DIST=1 BTC_SYNC_CHECK=0 GPU_MAX_HW_QUEUES=4;;
    direct_nocheck_q8) run $v BTC_BENCH_FORCE_DIST=1 BTC_SYNC_CHECK=0;;
    torch_q8) run $v BTC_BENCH_FORCE_DIST=1 BTC_SYNC_TRANSPORT=torch;;
    torch_q4) run $v BTC_BENCH_FORCE_DIST=1 BTC_SYNC_TRANSPORT=torch GPU_MAX_HW_QUEUES=4;;
    split_direct_q8) run $v BTC_BENCH_FORCE_DIST=1 BTC_SCHEDULE=split;;
    split_direct_q4) run $v BTC_BENCH_FORCE_DIST=1 BTC_SCHEDULE=split GPU_MAX_HW_QUEUES=4;;
    split_torch_q8) run $v BTC_BENCH_FORCE_DIST=1 BTC_SCHEDULE=split BTC_SYNC_TRANSPORT=torch;;
  esac
done
