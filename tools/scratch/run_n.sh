cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in kb8 kb14; do
  [ $v = kb14 ] && cp btcdet_amd/libbtcdet_hip_kb14.so btcdet_amd/libbtcdet_hip.so
  echo == $v
  CB_WARM=12 timeout 600 python tools/conv_bench.py split 2>&1 | grep -E " 27 +(4|5|6|16) +(4|6|16|32) | 27 +32 +5 |sum over"
  timeout 900 python -m pytest tests/test_hip_core.py -q -x 2>&1 | tail -2
done
