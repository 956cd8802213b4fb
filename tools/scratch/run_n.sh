cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NARROW=1 timeout 300 python tools/wgrad_bench.py 2>&1 | grep -E "^ *[0-9]+ +[0-9]+ +27 +32 +5 "
AB_STEPS=80 bash tools/ab_env.sh 3 "narrow:BTC_X=0" "off:BTC_TUNE=22=1"
