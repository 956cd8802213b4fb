#!/bin/bash
# end-of-round sanity: smoke(), and bench.py launched the way the driver launches N > 1 (one rank here: world size 1 over RCCL)
out=gpurun_out/r5o; mkdir -p $out
cd /root/repo
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
BTC_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_dist1.json 2> $out/bench_dist1.err; grep '^{' $out/bench_dist1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('world-1 over RCCL: %.1f scenes/s' % d['value'], d['config']['collective'], d['config'].get('predicted_scaling_eff'), d['roofline']['traffic'])"; tail -2 $out/bench_dist1.err
