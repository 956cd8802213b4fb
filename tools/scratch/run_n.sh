cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_wgrad_n.py -q 2>&1 | tail -4
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -E " (4|6) +16 |totals| 32 +5 "
AB_STEPS=80 bash tools/ab_env.sh 3 "narrow:BTC_X=0" "off:BTC_TUNE=22=1"
