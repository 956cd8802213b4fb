"""The item loops of the LDS-DMA apply kernels may wait for vector memory only where the source says so (the counted wait in front of
the item's barrier).  hipcc's waitcnt pass adds an `s_waitcnt vmcnt(0)` in front of any LDS read it cannot order against the LDS-DMA in
flight -- one whose memory operand carries no type information, e.g. a uint4 struct copy -- and such a wait between an item's barrier
and its last MFMA serialises the loads of item i + S - 1 with the products of item i (conv_apply_split.hip / conv_apply_bf16.hip carried
one in every instance until round 4).  Compile-only: runs wherever hipcc is, no GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "btcdet_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def loop_waits(asm, pattern):
    """{kernel: [(offset from the loop's barrier, wait)]} for every kernel whose name contains `pattern` and that has MFMAs"""
    lines = asm.split("\n")
    out = {}
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pattern in l]
    for i, name in starts:
        end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[i:end]
        mf = [j for j, l in enumerate(body) if "v_mfma" in l]
        if not mf:
            continue
        bar = max(j for j, l in enumerate(body[:mf[0]]) if "s_barrier" in l)
        out[name] = [(j - bar, body[j].strip()) for j in range(bar, mf[-1]) if "s_waitcnt vmcnt" in body[j]]
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,pattern,at_least", [("conv_apply_split.hip", "conv_apply_s", 20), ("conv_apply_bf16.hip", "conv_apply_b", 6)])
def test_no_vector_memory_wait_between_barrier_and_products(tmp_path, src, pattern, at_least):
    asm = tmp_path / "k.s"
    subprocess.run([HIPCC, "-S", "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off", "-I", CSRC,
                    os.path.join(CSRC, src), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    waits = loop_waits(asm.read_text(), pattern)
    assert len(waits) >= at_least, sorted(waits)
    bad = {k: v for k, v in waits.items() if v}
    assert not bad, "compiler-inserted vector-memory waits inside the product phase: %s" % bad
