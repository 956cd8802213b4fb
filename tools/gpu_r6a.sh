#!/bin/bash
# round 6, session a: the ADVICE fixes' tests, HEAD's numbers on this box (default schedule + in order, 80 steps), the CU-mask A/B of the
# weight-gradient stream, and one in-order step's kernels in launch order -> gpurun_out/r6a/
out=gpurun_out/r6a; mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_core.py tests/test_hip_sync_bn.py tests/test_hip_wgrad_x.py tests/test_hip_row_order.py tests/test_hip_split.py tests/test_hip_glue.py tests/test_hip_occupancy.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
b80() { env "$@" timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>> $out/bench.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['step_ms']['median'])"; }
echo "default:   $(b80 A=1)"
echo "in_order:  $(b80 BTC_SCHEDULE=in_order)"
for c in 64 128 192; do echo "wgrad CUs $c: default $(b80 BTC_WGRAD_CUS=$c)  in_order $(b80 BTC_WGRAD_CUS=$c BTC_SCHEDULE=in_order)"; done
echo "default again: $(b80 A=1)"
(cd /tmp && BTC_SCHEDULE=in_order BTC_DEFER_WGRAD=0 BTC_OVERLAP_MIN_ROWS=2000000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6a -o bench -- python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/alone.json 2> /root/repo/$out/alone.err)
find /tmp/prof_r6a -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/alone_kernel_stats.csv
find /tmp/prof_r6a -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/step_sequence.py {} 3 > $out/step_sequence.txt 2>&1
tail -1 $out/step_sequence.txt
timeout 300 python tools/op_sites.py > $out/op_sites.txt 2>&1; head -5 $out/op_sites.txt
