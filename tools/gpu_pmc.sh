#!/bin/bash
# HBM traffic of the kernel families from the PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# in SEPARATE passes (they do not fit one), kernel trace only -> gpurun_out/<tag>/pmc.json (copy to profiles/)
tag=${1:-r03_pmc}
extra=${2:-}      # further bench.py arguments of the passes ("--features bf16", "--workload waymo"); the output is pmc$3.json
sfx=${3:-}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}${sfx}_$c -o pmc -- python /root/repo/bench.py --priming 16 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras $extra > /root/repo/gpurun_out/$tag/bench${sfx}_$c.json 2> /root/repo/gpurun_out/$tag/bench${sfx}_$c.err
done
cd /root/repo
python tools/pmc_summary.py /tmp/pmc_${tag}${sfx}_FETCH_SIZE /tmp/pmc_${tag}${sfx}_WRITE_SIZE gpurun_out/$tag/pmc${sfx}.json "$extra"
