#!/bin/bash
out=gpurun_out/r6d; mkdir -p $out
cd /root/repo
b80() { env "$@" timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2> $out/last.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'])"; }
for i in 1 2 3 4 5 6; do echo "two ahead $(b80 A=1)   one ahead $(b80 BTC_BENCH_AHEAD=1)"; done
