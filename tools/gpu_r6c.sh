#!/bin/bash
# round 6, session c: the launch diet (one fill per voxelizer / occupancy targets, int32 twins, fused occupancy probability, cached seeds):
# full GPU suite, HEAD's numbers (default schedule + in order, 80 steps), one in-order step's kernels in launch order
out=gpurun_out/r6c; mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
b80() { env "$@" timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>> $out/bench.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['step_ms']['median'])"; }
echo "default:   $(b80 A=1)"
echo "in_order:  $(b80 BTC_SCHEDULE=in_order)"
echo "default:   $(b80 A=1)"
echo "in_order:  $(b80 BTC_SCHEDULE=in_order)"
(cd /tmp && BTC_SCHEDULE=in_order BTC_DEFER_WGRAD=0 BTC_OVERLAP_MIN_ROWS=2000000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6c -o bench -- python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/alone.json 2> /root/repo/$out/alone.err)
find /tmp/prof_r6c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/alone_kernel_stats.csv
find /tmp/prof_r6c -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/step_sequence.py {} 3 > $out/step_sequence.txt 2>&1
tail -1 $out/step_sequence.txt
