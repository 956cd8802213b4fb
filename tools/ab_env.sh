#!/bin/bash
# same-box A/B of bench.py under environment variants: tools/ab_env.sh REPEATS "NAME:VAR=VAL,VAR=VAL" ...   (NAME: alone = no variables)
mkdir -p gpurun_out/abenv
reps=$1; shift
for r in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%:*}; vars=${spec#*:}
    envs=$(echo "$vars" | tr ',' ' ')
    out=$(env $envs timeout 300 python bench.py --steps ${AB_STEPS:-100} --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | grep '^{' | tail -1)
    python - "$name" "$r" <<PY "$out"
import json, sys
try:
    d = json.loads(sys.argv[3])
    print("%-24s run %s  %7.1f scenes/s  %6.3f ms/step  median %.3f p90 %.3f" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["step_ms"]["median"], d["step_ms"]["p90"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
done
