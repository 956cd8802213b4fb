#!/bin/bash
# round 5: where the host threads of the pipelined step spend their time now (phases, cProfile of forward_det, all-thread sampler)
out=gpurun_out/r5h; mkdir -p $out
cd /root/repo
BTC_TRAINER_TIMING=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $out/timing.json 2> $out/timing.err; grep "trainer host" $out/timing.err; python -c "
import json; d=json.loads([l for l in open('$out/timing.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
timeout 300 python tools/host_profile_main.py 80 > $out/host_main.txt 2>&1; head -75 $out/host_main.txt | cut -c1-150
timeout 300 python tools/host_sampler.py > $out/host_sampler.txt 2>&1; grep -n "==== thread\|leaf" $out/host_sampler.txt | head; python - <<'PY'
import re
t = open("gpurun_out/r5h/host_sampler.txt").read()
for blk in t.split("==== thread")[1:]:
    name = blk.split("\n")[0]
    leaf = blk.split("-- leaf")[1] if "-- leaf" in blk else ""
    print("THREAD", name)
    print("\n".join(leaf.split("\n")[1:16]))
PY
