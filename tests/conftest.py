import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def exact_conv():
    """bit-exact comparisons with the oracle's fmaf chain: keep the fp32 launches off the split-operand kernel
    (csrc/conv_apply_split.hip, fp32-accurate but a different summation; its own bounds: tests/test_hip_split.py)"""
    from btcdet_amd._lib import check, lib
    check(lib().btc_tune_set(14, 1), "btc_tune_set")
    yield
    check(lib().btc_tune_set(14, 0), "btc_tune_set")
