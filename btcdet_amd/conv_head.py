"""The ROI head's pooling stage on the HIP operator set: `ConvHead.roi_conv_pool` and its helpers of
/root/reference/btcdet/models/roi_heads/conv_head.py:209-379 (local conv grids, feature splatting, the pooling itself),
:394-425 (ROI grid points) and :509-610 (trilinear read-out of the sparse feature volume), with the constructor protocol,
attribute names and parameter shapes of the reference's class (:12-147) so that its checkpoints load key for key.

What runs where: every ROI (B x 128 in training) is cut into a GRID_SIZE (3 x 3 x 3) lattice of points; around each lattice point
 * `SA_rawpoints` / `SA_occpoints` (pointnet2_stack.StackSAModuleMSG -> csrc/pointnet2.hip ball query + grouping) pool the raw and
   the occupancy-completed points, in the ROI's own frame (POINT_ROT),
 * a micro-scene of PART_SCENE_SIZE / KER_SIZE = [2, 4, 12] cells is laid out in the ROI's orientation, filled by trilinear
   read-out of `multi_scale_3d_features['x_combine']`, and reduced to one 128-vector by three strided SparseConv3d blocks --
   B x 128 x 27 = 6912 micro-scenes in one sparse tensor (csrc/rulebook.hip / conv kernels, tests/test_hip_roi_microscenes.py);
the three sources are concatenated per lattice point and handed to the shared FC / cls / reg layers.

Training side: target assignment (ProposalTargetLayer) and the RCNN cls / reg / corner losses of RoIHeadTemplate live in
btcdet_amd/roi_targets.py (`assign_targets`, `get_loss`); `forward` takes the ROIs from `batch_dict['rois']` or from the proposal
step (dense_head.proposal_layer)."""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import spconv
from .backbones_3d import post_act_block
from .dense_head import ResidualCoder, proposal_layer
from .pointnet2_stack import StackSAModuleMSG


def rotate_z(points, angle):
    """points (R, P, 3) rotated about z by angle (R,): x' = x cos - y sin, y' = x sin + y cos (common_utils.py:34-56)"""
    c, s = torch.cos(angle), torch.sin(angle)
    x, y = points[..., 0], points[..., 1]
    return torch.stack([x * c[:, None] - y * s[:, None], x * s[:, None] + y * c[:, None], points[..., 2]], dim=-1)


def lattice_points(boxes, grid_zyx, dim_times=1.0):
    """cell-centre lattice of every box in its own frame (conv_head.py:406-425, the e2e=False branch): boxes (R, 7+) [x, y, z, dx, dy,
    dz, yaw, ...], grid_zyx = cells along (z, y, x) -> (local points (R, P, 3) xyz, lattice index (R, P, 3) as (z, y, x) floats), P in
    z-major order"""
    gz, gy, gx = (int(v) for v in grid_zyx)
    dev = boxes.device
    iz, iy, ix = torch.meshgrid(torch.arange(gz, device=dev), torch.arange(gy, device=dev), torch.arange(gx, device=dev), indexing="ij")
    idx = torch.stack([iz.reshape(-1), iy.reshape(-1), ix.reshape(-1)], dim=-1).float()          # (P, 3) z, y, x
    size = boxes[:, 3:6] * dim_times                                                              # (R, 3) dx, dy, dz
    cells = torch.tensor([gx, gy, gz], dtype=torch.float32, device=dev)
    xyz = (idx.flip(-1)[None] + 0.5) * size[:, None, :] / cells.view(1, 1, 3) - size[:, None, :] / 2
    return xyz, idx[None].expand(boxes.shape[0], -1, -1)


def world_lattice(boxes, grid_zyx, dim_times=1.0):
    """the same lattice in world coordinates: rotated by the box's yaw about z and moved to its centre (conv_head.py:394-405)"""
    local, idx = lattice_points(boxes, grid_zyx, dim_times)
    return rotate_z(local, boxes[:, 6]) + boxes[:, None, 0:3], idx


def trilinear_readout(x, batch_index, zyx):
    """x: SparseConvTensor; zyx (Q, 3) fractional cell coordinates (cell centres at integers); batch_index (Q,).  Trilinear
    interpolation of the densified volume with out-of-range corners contributing zero (common_utils.py:247-311, normalize=False:
    corner weights are the usual products of distances, corner indices are clamped only to keep the gather in range)."""
    vol = x.dense()                                                                               # (B, C, D, H, W)
    D, H, W = (int(v) for v in x.spatial_shape)
    lo = torch.floor(zyx)
    frac = zyx - lo
    lo = lo.long()
    out = None
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                cz, cy, cx = lo[:, 0] + dz, lo[:, 1] + dy, lo[:, 2] + dx
                inside = (cz >= 0) & (cz < D) & (cy >= 0) & (cy < H) & (cx >= 0) & (cx < W)
                w = (frac[:, 0] if dz else 1 - frac[:, 0]) * (frac[:, 1] if dy else 1 - frac[:, 1]) * (frac[:, 2] if dx else 1 - frac[:, 2])
                v = vol[batch_index, :, cz.clamp(0, D - 1), cy.clamp(0, H - 1), cx.clamp(0, W - 1)]
                term = v * (w.abs() * inside).unsqueeze(-1)
                out = term if out is None else out + term
    return out


class _TrilinearRead(torch.autograd.Function):
    """out[p] = sum_c w[p][c] * feat[rows[p][c]] over the 8 corners of kept lattice point p (csrc/roi_pool.hip trilinear_gather); backward:
    the (point, corner) pairs stably sorted by the row they read, one wave per row summing in that order (trilinear_scatter) --
    deterministic, no float atomics, and only over the kept points (the reference's index_put(accumulate) runs over all of them)"""

    @staticmethod
    def forward(ctx, feat, rows_k, w_k):
        from ._lib import check, lib, ptr, stream_ptr
        feat = feat.contiguous()
        M, C = int(rows_k.shape[0]), int(feat.shape[1])
        out = torch.empty((M, C), dtype=torch.float32, device=feat.device)
        check(lib().btc_trilinear_gather(ptr(feat), C, M, ptr(rows_k), ptr(w_k), ptr(out), stream_ptr()), "btc_trilinear_gather")
        ctx.save_for_backward(rows_k, w_k)
        ctx.n_rows = int(feat.shape[0])
        return out

    @staticmethod
    def backward(ctx, grad):
        from ._lib import check, lib, ptr, stream_ptr
        rows_k, w_k = ctx.saved_tensors
        n_rows, C = ctx.n_rows, int(grad.shape[1])
        grad = grad.contiguous()
        key = torch.where((rows_k >= 0) & (w_k != 0), rows_k, torch.full_like(rows_k, n_rows)).view(-1)
        skey, perm = torch.sort(key, stable=True)                       # pairs e = 8 p + c grouped by row, in pair order inside a row
        seg = torch.searchsorted(skey, torch.arange(n_rows + 1, dtype=skey.dtype, device=skey.device)).int()
        gfeat = torch.empty((n_rows, C), dtype=torch.float32, device=grad.device)
        check(lib().btc_trilinear_scatter(ptr(grad), C, ptr(perm.int()), ptr(seg), ptr(w_k), n_rows, ptr(gfeat), stream_ptr()), "btc_trilinear_scatter")
        return gfeat, None, None


def trilinear_splat_resident(x, flat_points, points_per_batch, batch_size, point_cloud_range, voxel_size, stride_zyx):
    """-> (keep (M,) int64 lattice points that read something, features (M, C)) for the sparse tensor x at world points flat_points (Q, 3):
    trilinear_readout() + the non-zero filter of ConvHead.splat without x.dense() and without the (Q, C) intermediates -- the corner
    rows / weights of every point from ONE launch (btc_trilinear_corners), the read-out itself over the kept points only"""
    import ctypes
    from ._lib import check, lib, ptr, stream_ptr
    feats = x.features
    dev = feats.device
    D, H, W = (int(v) for v in x.spatial_shape)
    N = int(feats.shape[0])
    idx = x.indices.long()
    live = (feats.detach() != 0).any(dim=1).to(torch.uint8)          # (an all-zero row is read like any other but keeps no point alive)
    cell_row = torch.full((int(batch_size), D, H, W), -1, dtype=torch.int32, device=dev)
    cell_row[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = torch.arange(N, dtype=torch.int32, device=dev)
    Q = int(flat_points.shape[0])
    rows = torch.empty((Q, 8), dtype=torch.int32, device=dev)
    wts = torch.empty((Q, 8), dtype=torch.float32, device=dev)
    flag = torch.empty((Q,), dtype=torch.uint8, device=dev)
    f3, i3 = ctypes.c_float * 3, ctypes.c_int32 * 3
    check(lib().btc_trilinear_corners(ptr(flat_points), Q, int(points_per_batch), f3(*[float(v) for v in point_cloud_range[:3]]),
                                      f3(*[float(v) for v in voxel_size[:3]]), f3(*[float(v) for v in stride_zyx]), i3(D, H, W), int(batch_size),
                                      ptr(cell_row), ptr(live), ptr(rows), ptr(wts), ptr(flag), stream_ptr()), "btc_trilinear_corners")
    keep = torch.nonzero(flag)[:, 0]                                   # (the one read-back, as the reference's nonzero)
    return keep, _TrilinearRead.apply(feats, rows[keep].contiguous(), wts[keep].contiguous())


RESIDENT_SPLAT = True   # the read-out of x_combine without densifying it (csrc/roi_pool.hip); False: the dense formulation (tests compare the two)


class ConvHead(nn.Module):
    def __init__(self, input_channels, model_cfg, num_class=1, **kwargs):
        super().__init__()
        self.model_cfg, self.num_class = model_cfg, num_class
        pool = model_cfg.CONV_GRID_POOL
        coder = model_cfg.TARGET_CONFIG.BOX_CODER
        assert coder == "ResidualCoder", "the configured box coder only"
        self.box_coder = ResidualCoder(**model_cfg.TARGET_CONFIG.get("BOX_CODER_CONFIG", {}))
        self.fix_dims = model_cfg.TARGET_CONFIG.get("BOX_CODER_CONFIG", None) == "AbsResidualCoder"
        g = pool.GRID_SIZE
        self.grid_size = list(g) if isinstance(g, (list, tuple)) else [g, g, g]
        self.grid_num = int(np.prod(self.grid_size))
        self.dim_times = pool.get("DIM_TIMES", 1.0)
        self.point_rot, self.point_scale = pool.get("POINT_ROT", False), pool.get("POINT_SCALE", False)
        self.intrp_norm = pool.get("INTRP_NORM", False)
        assert not self.intrp_norm and not pool.get("VIS", False), "INTRP_NORM / VIS are not on the configured path"
        self.sources = list(pool.FEATURES_SOURCE)
        self.det_voxel_size, self.point_cloud_range = kwargs["det_voxel_size"], kwargs["point_cloud_range"]
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        layer_cfg = pool.CONV_LAYER
        self.raw_points_agg = self.occ_points_agg = None
        c_out = 0
        for src, attr, cin in (("raw_points", "SA_rawpoints", kwargs.get("num_rawpoint_features", 4) - 3), ("occ_points", "SA_occpoints", 1)):
            if src not in self.sources:
                continue
            mlps = [[cin] + list(m) for m in layer_cfg[src].MLPS]
            setattr(self, attr, StackSAModuleMSG(radii=layer_cfg[src].POOL_RADIUS, nsamples=layer_cfg[src].NSAMPLE, mlps=mlps, use_xyz=True,
                                                 pool_method="max_pool"))
            agg = layer_cfg[src].get("AGG_MLPS", None)
            if agg is not None:
                seq = []
                for a, b in zip(agg[:-1], agg[1:]):
                    seq += [nn.Conv2d(a, b, kernel_size=1, bias=False), nn.BatchNorm2d(b), nn.ReLU()]
                setattr(self, src + "_agg", nn.Sequential(*seq))
                c_out += agg[-1]
            else:
                c_out += sum(m[-1] for m in mlps)
        self.conv_layers, self.conv_layer_names, self.size_map, self.strides = nn.ModuleList(), [], {}, {}
        for src in self.sources:
            if src in ("bev_conv", "raw_points", "occ_points"):
                continue
            c = layer_cfg[src]
            scene = np.array(c.PART_SCENE_SIZE, dtype=np.float64)                                 # zyx lo, zyx hi
            assert len(scene) == 6, "3-D feature sources only (the bev source is not on the configured path)"
            local = np.around((scene[3:] - scene[:3]) / np.array(c.KER_SIZE, dtype=np.float64)).astype(int)
            self.size_map[src] = {"local_grid_size": local, "scene_times": c.get("SCENE_TIMES", 1),
                                  "dims": [float(scene[i + 3] - scene[i]) for i in (2, 1, 0)]}
            self.strides[src] = c.DOWNSAMPLE_FACTOR
            ch = c.CHANNEL
            self.conv_layers.append(spconv.SparseSequential(*[
                post_act_block(ch[i], ch[i + 1], c.KERNEL[i], norm_fn=norm_fn, stride=c.STRIDE[i], padding=c.PADDING[i],
                               indice_key="%s_spconv%d" % (src, i), conv_type="spconv") for i in range(len(c.STRIDE))]))
            self.conv_layer_names.append(src)
            c_out += ch[-1]
        self.pooled_channels = c_out
        pre = self.grid_num * c_out
        conv3d = model_cfg.get("SHARED_3D_CONV", None)
        if conv3d is not None and len(conv3d.KERNEL) > 0:
            seq, cur = [], c_out
            for k in range(len(conv3d.KERNEL)):
                seq += [nn.Conv3d(cur, conv3d.CHANNEL[k], kernel_size=conv3d.KERNEL[k], bias=False, stride=conv3d.STRIDE[k], padding=conv3d.PADDING[k]),
                        nn.BatchNorm3d(conv3d.CHANNEL[k]), nn.ReLU()]
                cur = conv3d.CHANNEL[k]
            self.shared_3dconv_layer = nn.Sequential(*seq)
            pre = cur
        fc = model_cfg.get("SHARED_FC", None)
        if fc is not None and len(fc) > 0:
            seq = []
            for k, width in enumerate(fc):
                seq += [nn.Conv1d(pre, width, kernel_size=1, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
                pre = width
                if k != len(fc) - 1 and model_cfg.DP_RATIO > 0:
                    seq.append(nn.Dropout(model_cfg.DP_RATIO))
            self.shared_fc_layer = nn.Sequential(*seq)
        self.cls_layers = self._fc(pre, num_class, model_cfg.CLS_FC)
        self.reg_layers = self._fc(pre, self.box_coder.code_size * num_class, model_cfg.REG_FC)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)
        self.forward_ret_dict = None

    def _fc(self, cin, cout, widths):
        """roi_head_template.py:30-44: Conv1d / BatchNorm1d / ReLU per width, dropout behind the first, a biased Conv1d last"""
        seq = []
        for k, w in enumerate(widths):
            seq += [nn.Conv1d(cin, w, kernel_size=1, bias=False), nn.BatchNorm1d(w), nn.ReLU()]
            cin = w
            if self.model_cfg.DP_RATIO >= 0 and k == 0:
                seq.append(nn.Dropout(self.model_cfg.DP_RATIO))
        seq.append(nn.Conv1d(cin, cout, kernel_size=1, bias=True))
        return nn.Sequential(*seq)

    # ------------------------------------------------------------------------------------------------------ the pooling stage
    def micro_scene_points(self, lattice, rois, local_grid, scene_times):
        """conv_head.py:209-222: every lattice point of every ROI becomes the centre of a box with the ROI's size (x scene_times) and
        yaw, and that box gets a `local_grid` (z, y, x) lattice of its own -> (world points (R*G, P, 3), their (z, y, x) lattice index)"""
        R = rois.shape[0]
        boxes = rois[:, None, :].repeat(1, self.grid_num, 1).view(R * self.grid_num, -1).clone()
        boxes[:, 0:3] = lattice.reshape(-1, 3)
        boxes[:, 3:6] = boxes[:, 3:6] * scene_times
        return world_lattice(boxes, local_grid, 1.0)

    def splat(self, x, points, lattice_idx, batch_size, stride):
        """conv_head.py:236-246, 544-566: trilinear read-out of sparse tensor `x` at the micro-scenes' points; points that read nothing
        but zeros are dropped -> (coords (M, 4) [micro-scene, z, y, x] int32, features (M, C))"""
        S, P, _ = points.shape
        per_scene = S // batch_size
        s = [stride] * 3 if isinstance(stride, int) else list(stride)
        rng, vs = self.point_cloud_range, self.det_voxel_size
        flat = points.reshape(-1, 3)
        if RESIDENT_SPLAT and x.features.is_cuda and x.features.dtype == torch.float32:
            keep, feat = trilinear_splat_resident(x, flat.contiguous().float(), P * per_scene, batch_size, rng, vs, s)
            coords = torch.cat([(keep // P).view(-1, 1), lattice_idx.reshape(-1, 3).long()[keep]], dim=-1)
            return coords.int(), feat
        zyx = torch.stack([(flat[:, 2] - float(rng[2])) / float(vs[2]) / s[0] - 0.5, (flat[:, 1] - float(rng[1])) / float(vs[1]) / s[1] - 0.5,
                           (flat[:, 0] - float(rng[0])) / float(vs[0]) / s[2] - 0.5], dim=-1)
        scene = torch.arange(S, device=points.device)
        b = (scene // per_scene)[:, None].expand(S, P).reshape(-1)
        feat = trilinear_readout(x, b, zyx)
        keep = torch.nonzero((feat.abs() > 0).any(dim=-1))[:, 0]
        coords = torch.cat([scene[:, None].expand(S, P).reshape(-1, 1), lattice_idx.reshape(-1, 3).long()], dim=-1)
        return coords[keep].int(), feat[keep]

    def roi_conv_pool(self, batch_dict):
        """conv_head.py:247-379 -> (pooled (B*N, grid_num * C, 1), {}); C = the sources' channels concatenated per lattice point"""
        B = batch_dict["batch_size"]
        rois = batch_dict["rois"]
        N = rois.shape[1]
        flat = rois.reshape(B * N, -1)
        lattice, _ = world_lattice(flat, self.grid_size, self.dim_times)                         # (B*N, G, 3)
        centres = lattice.reshape(-1, 3).contiguous()
        from .pointnet2_stack import with_host_counts
        centre_cnt = with_host_counts(torch.full((B,), N * self.grid_num, dtype=torch.int32, device=rois.device), [N * self.grid_num] * B)
        rot = xy = zs = None
        if self.point_rot:                                                                        # rotation into the ROI's frame
            c, s = torch.cos(-flat[:, 6]), torch.sin(-flat[:, 6])
            o, l = torch.zeros_like(c), torch.ones_like(c)
            rot = torch.stack([c, -s, o, s, c, o, o, o, l], dim=-1).view(-1, 3, 3)
        if self.point_scale:
            xy = torch.sqrt(flat[:, 3] ** 2 + flat[:, 4] ** 2).view(-1, 1, 1, 1).repeat(1, self.grid_num, 1, 1).view(-1, 1, 1)
            zs = flat[:, 5].view(-1, 1, 1, 1).repeat(1, self.grid_num, 1, 1).view(-1, 1, 1)
        parts = []
        for src, attr, agg in (("raw_points", "SA_rawpoints", self.raw_points_agg), ("occ_points", "SA_occpoints", self.occ_points_agg)):
            if src not in self.sources:
                continue
            if src == "raw_points":
                pts = batch_dict["points"]
                xyz, feats, bidx = pts[:, 1:4], (pts[:, 4:].contiguous() if pts.shape[1] > 4 else None), pts[:, 0]
            else:
                pts = batch_dict["occ_pnts"]
                xyz, feats, bidx = pts[:, 0:3], (pts[:, 3:].contiguous() if pts.shape[1] > 3 else None), batch_dict["added_occ_b_ind"]
            cnt = torch.bincount(bidx.long(), minlength=B)[:B].int()
            _, pooled = getattr(self, attr)(xyz=xyz.contiguous(), xyz_batch_cnt=cnt, new_xyz=centres, new_xyz_batch_cnt=centre_cnt, features=feats,
                                            rotateMatrix=rot, xyscales=xy, zscales=zs)
            if agg is not None:
                pooled = agg(pooled.view(B * N * self.grid_num, -1, 1, 1))
            parts.append(pooled.view(B * N * self.grid_num, -1))
        for src, layer in zip(self.conv_layer_names, self.conv_layers):
            sm = self.size_map[src]
            points, lat_idx = self.micro_scene_points(lattice, flat, sm["local_grid_size"], sm["scene_times"])
            feats_in = batch_dict["multi_scale_3d_features"][src]
            coords, feats = self.splat(feats_in, points, lat_idx, B, self.strides[src])
            scenes = spconv.SparseConvTensor(features=feats, indices=coords, spatial_shape=[int(v) for v in sm["local_grid_size"]],
                                             batch_size=B * N * self.grid_num)
            parts.append(torch.squeeze(layer(scenes).dense()))
        out = torch.cat(parts, dim=-1) if len(parts) > 1 else parts[0]
        out = out.view(B * N, *self.grid_size, out.shape[-1]).permute(0, 4, 1, 2, 3).contiguous()     # (BN, C, gz, gy, gx)
        if getattr(self, "shared_3dconv_layer", None) is not None:
            out = self.shared_3dconv_layer(out)
        return out.view(B * N, -1, 1), {}

    # ---------------------------------------------------------------------------------------------------------------- forward
    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        """conv_head.py:428-457: residuals decoded in the ROI's frame, rotated back and moved to its centre"""
        code = self.box_coder.code_size
        local = rois.clone().detach()
        local[:, :, 0:3] = 0
        boxes = self.box_coder.decode_torch(box_preds.view(batch_size, -1, code), local).view(-1, code)
        xyz = rotate_z(boxes[:, None, 0:3], rois[:, :, 6].reshape(-1))[:, 0]
        boxes = torch.cat([xyz + rois[:, :, 0:3].reshape(-1, 3), boxes[:, 3:]], dim=-1)
        return cls_preds.view(batch_size, -1, cls_preds.shape[-1]), boxes.view(batch_size, -1, code)

    def assign_targets(self, batch_dict, sampled_inds=None):
        """RoIHeadTemplate.assign_targets (roi_head_template.py:102-134): ROI_PER_IMAGE sampled rois per scene with their matched boxes
        in the roi's frame (btcdet_amd/roi_targets.py: resident, no read-back)"""
        from .roi_targets import ProposalTargetLayer, canonical_targets
        if getattr(self, "proposal_target_layer", None) is None:
            self.proposal_target_layer = ProposalTargetLayer(self.model_cfg.TARGET_CONFIG)
        return canonical_targets(self.proposal_target_layer(batch_dict, sampled_inds))

    def get_loss(self, tb_dict=None):
        """RoIHeadTemplate.get_loss (roi_head_template.py:222-232): rcnn_loss = cls + reg (+ corner); the logged scalars are copied to
        pinned memory asynchronously (LazyScalar, as OccHead3D's) instead of five `.item()` stalls"""
        from .occ_head import _lazy_scalars
        from .roi_targets import rcnn_cls_loss, rcnn_reg_loss
        tb_dict = {} if tb_dict is None else tb_dict
        ret, cfg = self.forward_ret_dict, self.model_cfg.LOSS_CONFIG
        loss_cls = rcnn_cls_loss(ret["rcnn_cls"], ret["rcnn_cls_labels"], cfg)
        loss_reg, corner = rcnn_reg_loss(ret, self.box_coder, cfg)
        loss = loss_cls + loss_reg
        names = ["rcnn_loss_cls", "rcnn_loss_reg", "rcnn_loss"] + (["rcnn_loss_corner"] if corner is not None else [])
        vals = [loss_cls, loss_reg, loss] + ([corner] if corner is not None else [])
        tb_dict.update(zip(names, _lazy_scalars(torch.stack([v.detach().float() for v in vals]), len(vals))))
        return loss, tb_dict

    def forward(self, batch_dict):
        targets = None
        if "batch_box_preds" in batch_dict:     # behind a dense head: its proposals (a caller may hand `rois` over instead)
            proposal_layer(batch_dict, self.model_cfg.NMS_CONFIG["TRAIN" if self.training else "TEST"])
        if self.training and "gt_boxes" in batch_dict:
            targets = self.assign_targets(batch_dict)
            batch_dict["rois"], batch_dict["roi_labels"] = targets["rois"], targets["roi_labels"]
        pooled, _ = self.roi_conv_pool(batch_dict)
        batch_dict["pooled_features"] = pooled
        shared = self.shared_fc_layer(pooled) if getattr(self, "shared_fc_layer", None) is not None else pooled
        rcnn_cls = self.cls_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        rcnn_reg = self.reg_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        if self.training:
            self.forward_ret_dict = dict(targets or {"rois": batch_dict["rois"]}, rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg)
        else:
            batch_dict["batch_cls_preds"], batch_dict["batch_box_preds"] = self.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"],
                                                                                                         rcnn_cls, rcnn_reg)
            batch_dict["cls_preds_normalized"] = False
        return batch_dict


__all__ = {"ConvHead": ConvHead}
