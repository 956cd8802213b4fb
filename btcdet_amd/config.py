"""Attribute-style config dict (the reference uses EasyDict, /root/reference/btcdet/config.py:51-81;
modules read ``cfg.KEY`` and ``cfg.get('KEY', default)``)."""
import os

import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_attr(v) for v in o]
    return o


def load_cfg(path=None):
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs", "btcdet_kitti_car.yaml")
    with open(path) as f:
        return to_attr(yaml.safe_load(f))
