#!/usr/bin/env python3
"""Which `s_waitcnt vmcnt(N)` sit INSIDE the item loop of the LDS-DMA apply kernels (between the loop's s_barrier and its last MFMA)?
The counted waits written in the source are the only ones that should be there: a compiler-inserted vmcnt(0) in front of a ds_read
(SIInsertWaitcnts orders LDS reads behind every LDS-DMA in flight it cannot prove disjoint) serialises load and compute of every item.
    hipcc -S --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -ffp-contract=off -I btcdet_amd/csrc btcdet_amd/csrc/conv_apply_split.hip -o /tmp/split.s
    python tools/isa_waits.py /tmp/split.s conv_apply_s"""
import re
import subprocess
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l]
for i, name in starts:
    end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
    body = lines[i:end]
    mf = [j for j, l in enumerate(body) if "v_mfma" in l]
    if not mf:
        continue
    bar = max(j for j, l in enumerate(body[:mf[0]]) if "s_barrier" in l)
    inside = [(j, body[j].strip()) for j in range(bar - 4, mf[-1]) if "s_waitcnt vmcnt" in body[j]]
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        dem = name
    print("%-70s mfma %3d  waits in loop: %s" % (dem[-70:], len(mf), ", ".join("%s@+%d" % (w.split()[-1], j - bar) for j, w in inside)))
