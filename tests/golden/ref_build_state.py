"""Builds the REFERENCE's own BtcNet (btcdet/models/detectors/btcnet.py, its Detector3DTemplate.build_networks and its
tools/cfgs/model_configs/btcdet_kitti_car.yaml) on CPU with `spconv` resolved to btcdet_amd.spconv, and prints the
state_dict keys and shapes as JSON.  Runs only where /root/reference is mounted (this container).  Used by
tests/test_reference_dropin_cpu.py and to (re)generate tests/golden/ref_state_keys.json:

    python tests/golden/ref_build_state.py > tests/golden/ref_state_keys.json

Stubs: easydict, skimage, the reference's compiled *_cuda extension modules (absent); device="cuda"
literals in torch factory calls are rewritten to "cpu".  Nothing of the reference is copied: its modules are imported."""
import os, sys, types, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); REF = "/root/reference"
sys.path.insert(0, ROOT); sys.path.insert(0, REF)
import scipy.spatial
sys.stdout = sys.stderr  # the reference prints while it builds; only the final JSON line goes to stdout
import torch
np.int=int; np.float=float
def _wrap(fn):
    def w(*a, **k):
        if "device" in k and (k["device"]=="cuda" or str(k["device"]).startswith("cuda")): k["device"]="cpu"
        return fn(*a, **k)
    return w
for name in ["zeros","ones","tensor","as_tensor","arange","zeros_like","ones_like","rand","randint","empty","full"]:
    setattr(torch, name, _wrap(getattr(torch, name)))
torch.Tensor.cuda = lambda self,*a,**k: self
torch.nn.Module.cuda = lambda self,*a,**k: self
def _mod(name, **attrs):
    m=types.ModuleType(name)
    for k,v in attrs.items(): setattr(m,k,v)
    sys.modules[name]=m; return m
import btcdet_amd
btcdet_amd.install_as_spconv()
class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items(): self[k] = v
    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict): v = _EasyDict(v)
        elif isinstance(v, list): v = [_EasyDict(x) if isinstance(x, dict) else x for x in v]
        super().__setitem__(k, v)
    __setattr__ = __setitem__
    def __getattr__(self, k):
        try: return self[k]
        except KeyError: raise AttributeError(k)
_mod("easydict", EasyDict=_EasyDict)
_mod("skimage"); _mod("skimage.draw", line_aa=None); _mod("skimage.io")
from btcdet_amd import iou3d_nms as _amd_nms
_amd_nms.install_as_iou3d_nms_cuda()  # the reference's iou3d_nms_utils.py binds to this stand-in for its compiled module
for n in ["btcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda","btcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda","btcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda"]:
    _mod(n)
import yaml
cfg = yaml.safe_load(open(os.path.join(REF,"tools/cfgs/model_configs/btcdet_kitti_car.yaml")))
base = yaml.safe_load(open(os.path.join(REF,"tools",cfg["DATA_CONFIG"]["_BASE_CONFIG_"])))
data=dict(base); data.update({k:v for k,v in cfg["DATA_CONFIG"].items() if k!="_BASE_CONFIG_"}); cfg["DATA_CONFIG"]=data
cfg=_EasyDict(cfg)
from btcdet.models.detectors.btcnet import BtcNet
from btcdet.datasets.processor.data_processor import DataProcessor
class DS: pass
ds=DS(); d=cfg.DATA_CONFIG
ds.dataset_cfg=d; ds.class_names=cfg.CLASS_NAMES; ds.training=True; ds.mode='train'
ds.point_cloud_range=np.array(d.POINT_CLOUD_RANGE,dtype=np.float32); ds.occ_point_cloud_range=np.array(d.OCC.POINT_CLOUD_RANGE,dtype=np.float32)
ds.data_processor=DataProcessor(d.DATA_PROCESSOR, point_cloud_range=ds.occ_point_cloud_range, training=True, occ_config=d.OCC, det_point_cloud_range=ds.point_cloud_range)
for a in ("occ_grid_size","occ_voxel_size","det_grid_size","det_voxel_size","occ_dim"): setattr(ds,a,getattr(ds.data_processor,a))
ds.grid_size=ds.det_grid_size; ds.voxel_size=ds.det_voxel_size
class PFE: 
    num_point_features=len(d.POINT_FEATURE_ENCODING.used_feature_list)
ds.point_feature_encoder=PFE()
ds.depth_downsample_factor=None
net=BtcNet(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=ds, full_config=cfg)
sd={k:list(v.shape) for k,v in net.state_dict().items()}
sys.stdout = sys.__stdout__
from btcdet.ops.iou3d_nms import iou3d_nms_utils as _ref_nms_utils
_nms_bound = (_ref_nms_utils.iou3d_nms_cuda is _amd_nms.iou3d_nms_cuda and
              all(hasattr(_ref_nms_utils.iou3d_nms_cuda, f) for f in ("boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu", "boxes_iou_bev_cpu")))
print(json.dumps({"n": len(sd), "keys": sd, "iou3d_nms_utils_bound_to_btcdet_amd": bool(_nms_bound)}))
