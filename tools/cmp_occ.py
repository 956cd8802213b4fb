"""debug helper: compare a GPU dump (tools/dump_occ.py) with the CPU oracle"""
import sys, numpy as np, torch, warnings
warnings.filterwarnings("ignore")
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
from golden_batch import golden_batch
from oracle import occ_oracle
from btcdet_amd.config import load_cfg
import torch.nn.functional as F
cfg=load_cfg(); g,sc,bd=golden_batch(); O=occ_oracle.OccOracle(cfg); ref=O.targets(bd)
d=np.load('/root/repo/gpurun_out/occ_dump_golden.npz')
shape=(2,9,157,209)
def um(k): return torch.from_numpy(np.unpackbits(d[k])[:np.prod(shape)].reshape(shape).astype(bool))
for k in ["occ_voxelwise_mask","general_cls_loss_mask","fore_voxelwise_mask","bm_voxelwise_mask","forebox_label","pos_mask","occ_mirr_cls_mask","occ_bm_cls_mask","occ_fore_cls_mask","general_reg_loss_mask"]:
    a=um(k); b=ref[k]>0
    print("%-28s gpu %7d ref %7d diff %6d (gpu-only %d ref-only %d)"%(k,a.sum(),b.sum(),(a!=b).sum(),(a&~b).sum(),(~a&b).sum()))
print("pos_all_num", int(d["pos_all_num"]), int(ref["pos_all_num"]))
