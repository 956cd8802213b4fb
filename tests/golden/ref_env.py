"""Makes the REAL reference (/root/reference, Python) importable in this container: stubs for the absent
third-party / compiled modules, the removed NumPy aliases, and device="cuda" -> "cpu" in torch factory calls.  Used by the
golden-vector generators only; nothing here travels to the GPU box and nothing of the reference is copied.

    install(spconv_module)   spconv_module: what `import spconv` resolves to inside the reference's files
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, list):
            v = [EasyDict(x) if isinstance(x, dict) else x for x in v]
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install(spconv_module):
    for p in (ROOT, REF, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import scipy.spatial  # noqa: F401  (must precede the np.int alias)
    import torch
    np.int = int
    np.float = float

    def wrap(fn):
        def w(*a, **k):
            if "device" in k and (k["device"] == "cuda" or str(k["device"]).startswith("cuda")):
                k["device"] = "cpu"
            return fn(*a, **k)
        return w

    if not getattr(torch, "_btc_ref_env", False):
        for name in ["zeros", "ones", "tensor", "as_tensor", "arange", "zeros_like", "ones_like", "rand", "randint", "empty", "full"]:
            setattr(torch, name, wrap(getattr(torch, name)))
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch._btc_ref_env = True
    sys.modules["spconv"] = spconv_module
    sys.modules["spconv.utils"] = spconv_module.utils
    _mod("easydict", EasyDict=EasyDict)
    _mod("skimage")
    _mod("skimage.draw", line_aa=None)
    _mod("skimage.io")
    for n in ["btcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda", "btcdet.ops.iou3d_nms.iou3d_nms_cuda",
              "btcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda", "btcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda"]:
        _mod(n)


def load_ref_cfg():
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REF, "tools/cfgs/model_configs/btcdet_kitti_car.yaml")))
    base = yaml.safe_load(open(os.path.join(REF, "tools", cfg["DATA_CONFIG"]["_BASE_CONFIG_"])))
    data = dict(base)
    data.update({k: v for k, v in cfg["DATA_CONFIG"].items() if k != "_BASE_CONFIG_"})  # config.py:51-68 merge
    cfg["DATA_CONFIG"] = data
    return EasyDict(cfg)
