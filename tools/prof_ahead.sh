export TMPDIR=/tmp
cd /tmp && BTC_HANDOVER_WORKER=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ah -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
find /tmp/prof_ah -name "*kernel_trace.csv" | head -1 | xargs -I{} python /root/repo/tools/stream_timeline.py {} | cut -c1-600
