"""OccTargets3D: occupancy / occlusion training-target generator on the GPU.

Mirror of /root/reference/btcdet/models/occ_pnt/occ_training_targets/occ_targets_3d.py:8-171 and
occ_targets_template.py:11-469 (constructor protocol of detector3d_template.py:116-130, module
protocol ``forward(batch_dict) -> batch_dict``, batch_dict keys of SURVEY.md App. E) for the
configured variant: COORD_TYPE cylinder, REG True, TMPLT True, DROPOUT_RATE 0.  The reference's
chain of torch ops + python loop over the batch + ~17 host syncs is one C-ABI call here
(btc_occ_targets, csrc/occupancy.hip)."""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from ._lib import BtcOccBuffers, BtcOccConfig, OCC_BUFFER_FIELDS, check, lib, ptr, stream_ptr, workspace


def cylinder_voxel_centers(grid_size, occ_range, voxel_size, device):
    """all_voxel_centers (nz,ny,nx,3) and its BEV mean (ny*nx,2): Detector3DTemplate.create_subvox_loc
    (detector3d_template.py:52-63) for COORD_TYPE cylinder, in float32 like the reference."""
    nx, ny, nz = [int(g) for g in grid_size]
    vs = torch.tensor([voxel_size[2], voxel_size[1], voxel_size[0]], dtype=torch.float32)
    org = torch.tensor([occ_range[2], occ_range[1], occ_range[0]], dtype=torch.float32)
    z, y, x = torch.meshgrid(torch.arange(nz), torch.arange(ny), torch.arange(nx), indexing="ij")
    c = (0.5 + torch.stack([z, y, x], dim=0).to(torch.float32)) * vs.view(3, 1, 1, 1) + org.view(3, 1, 1, 1)
    rho, th, zz = c[2], c[1], c[0]
    ctr = torch.stack([rho * torch.cos(th * np.pi / 180.), -rho * torch.sin(th * np.pi / 180.), zz], dim=-1)
    return {"all_voxel_centers": ctr.to(device).contiguous(),
            "all_voxel_centers_2d": torch.mean(ctr[:, :, :, :2], dim=0).view(-1, 2).to(device)}


def backproject_table_host(sphere_grid, sphere_range, sphere_voxel, grid, occ_range, occ_voxel, sphere_offset=(0.0, 0.0, 0.0)):
    """(snz, sny, snx) int32 on the HOST: the cylinder cell z*ny*nx + y*nx + x that the corner of sphere cell [sz][sy][sx]
    back-projects into, -1 when it leaves the cylinder range -- occ_from_cylin_ocp's second half
    (/root/reference/btcdet/models/occ_pnt/occ_training_targets/occ_targets_template.py:146-152: index * reverse_sphere_voxel_size +
    rever_sphere_origin_tensor -> coords_utils.sphere_uvd2absxyz (:180-186) -> cartesian_cylinder_coords (:229-239) ->
    point2coords_inrange (:82-90)) evaluated ONCE for the whole corner lattice instead of per batch for the occluded cells: the
    lattice does not move with the batch.  Same torch ops in the same order and dtypes as the reference, on the host, so the table
    carries the host platform's float32 cos / sin / sqrt / atan2 -- the arithmetic a CPU run of the reference quantises on (every
    corner lies exactly ON an azimuth cell boundary, so the cell index is decided by the last ulp: torch-CPU = MKL VML for cos /
    sin / sqrt and Sleef_atan2f_u10, none of them correctly rounded; a CUDA run of the reference differs from both).  The kernels
    then look cells up instead of evaluating transcendentals (csrc/occupancy.hip occ_ray_project)."""
    snx, sny, snz = [int(v) for v in sphere_grid]
    nx, ny, nz = [int(v) for v in grid]
    rev_vs = torch.as_tensor([sphere_voxel[2], sphere_voxel[1], sphere_voxel[0]], dtype=torch.float32)
    rev_origin = torch.as_tensor([[sphere_range[2], sphere_range[1], sphere_range[0]]], dtype=torch.float32)
    z, y, x = torch.meshgrid(torch.arange(snz), torch.arange(sny), torch.arange(snx), indexing="ij")
    ind = torch.stack([z, y, x], dim=-1).view(-1, 3)                          # what torch.nonzero yields for a full mask
    sp = ind * rev_vs + rev_origin                                             # (el, az, r) of every corner, float32
    sx, sy, sz = sp[..., 2], sp[..., 1], sp[..., 0]
    xyd = sx * torch.cos(sz * np.pi / 180.)
    carte = torch.stack([xyd * torch.cos(sy * np.pi / 180.), -xyd * torch.sin(sy * np.pi / 180.), sx * torch.sin(sz * np.pi / 180.)], dim=-1)
    carte = carte - torch.as_tensor([list(sphere_offset)], dtype=torch.float32)
    sq = torch.square(carte)
    cyl = torch.stack([torch.sqrt(torch.sum(sq[..., 0:2], dim=-1)), torch.atan2(-carte[..., 1], carte[..., 0]) * (180. / np.pi), carte[..., 2]], dim=-1)
    origin = torch.as_tensor([list(occ_range[:3])], dtype=torch.float32)
    pmax = torch.as_tensor([list(occ_range[3:6])], dtype=torch.float32)
    vs = torch.as_tensor([list(occ_voxel)], dtype=torch.float32)
    ok = torch.cat([cyl >= origin, cyl <= pmax], dim=-1).all(-1)
    c = ((cyl - origin) / vs).to(torch.int64)
    c = torch.maximum(torch.minimum(c, torch.as_tensor([[nx - 1, ny - 1, nz - 1]], dtype=torch.int64)), torch.zeros((1, 3), dtype=torch.int64))
    lut = torch.where(ok, (c[..., 2] * ny + c[..., 1]) * nx + c[..., 0], torch.full_like(c[..., 0], -1))
    return lut.to(torch.int32).view(snz, sny, snx).contiguous()


class OccTargets3D(nn.Module):
    def __init__(self, model_cfg, voxel_size, point_cloud_range, data_cfg, grid_size, num_class, voxel_centers):
        super().__init__()
        self.model_cfg, self.data_cfg, self.num_class = model_cfg, data_cfg, num_class
        occ = data_cfg.OCC
        assert occ.COORD_TYPE == "cylinder", "only the configured cylinder occupancy grid is implemented"
        self.dropout_rate = float(occ.get("DROPOUT_RATE", 0.0))            # occ_targets_template.py:297-328 (round 4)
        self.dropout_rmv = bool(occ.get("DROPOUT_RMV", False))
        self.reverse_vis = model_cfg.PARAMS.get("REVERSE_VIS", "NOTHING")   # occ_targets_template.py:110-134 (round 4)
        assert self.reverse_vis in ("NOTHING", "VCC", "BACK_TRACK"), self.reverse_vis
        self.reg = model_cfg.PARAMS.get("REG", False)
        assert self.reg and model_cfg.TARGETS.TMPLT, "configured variant: REG True, TMPLT True"
        self.nx, self.ny, self.nz = [int(g) for g in grid_size]
        self.point_cloud_range = point_cloud_range
        self.all_voxel_centers = voxel_centers["all_voxel_centers"]
        self.all_voxel_centers_2d = voxel_centers["all_voxel_centers_2d"]
        # frozen 3x3 ones conv of the reference (occ_targets_template.py:30-32): kept so that
        # `occ_modules.occ_targets.fix_conv_2dzy.weight` exists in checkpoints
        self.fix_conv_2dzy = torch.nn.Conv2d(1, 1, kernel_size=3, stride=1, padding=1, bias=False)
        self.fix_conv_2dzy.weight.data.fill_(1.0)
        self.fix_conv_2dzy.requires_grad_(False)
        sr = np.asarray(occ.SUPPORT_SPHERE_RANGE, dtype=np.float64)
        if hasattr(occ, 'SUPPORT_SPHERE_VOXEL_SIZE'):
            svs = np.array([occ.SUPPORT_SPHERE_VOXEL_SIZE[0], occ.SUPPORT_SPHERE_VOXEL_SIZE[1], sr[6]])
        else:
            svs = np.array([voxel_size[0], voxel_size[1], sr[6]])
        sgrid = ((sr[3:6] - sr[:3]) / svs).astype(int)  # truncation, occ_targets_template.py:51
        self.sphere_nx, self.sphere_ny, self.sphere_nz = [int(v) for v in sgrid]
        lw = model_cfg.OCC_DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS
        kern = occ.DIST_KERN
        self.concede_x = occ.get("CONCEDE_X", kern[-1] // 2 if occ.get("HALF_X", False) else 0)
        self.point_coding = occ.get("USE_ABSXYZ", "original")
        assert self.point_coding is True or self.point_coding == "absxyz", "configured variant: USE_ABSXYZ True"
        c = BtcOccConfig()
        c.grid[:] = [self.nx, self.ny, self.nz]
        c.sphere_grid[:] = [self.sphere_nx, self.sphere_ny, self.sphere_nz]
        c.dist_kern[:] = [int(k) for k in kern]
        c.concede_x = int(self.concede_x)
        et = occ.EMPT_SUR_THRESH
        c.empt_sur_thresh = int(et) if (et != "None" and et < 9) else -1
        c.use_box_weight = int(occ.BOX_WEIGHT != 1.0)
        c.occ_range[:] = [float(np.float32(v)) for v in point_cloud_range]
        c.occ_voxel[:] = [float(np.float32(v)) for v in voxel_size]
        c.sphere_range[:] = [float(np.float32(v)) for v in sr[:6]]
        c.sphere_voxel[:] = [float(np.float32(v)) for v in svs]
        c.det_zmin, c.det_zmax = float(data_cfg.POINT_CLOUD_RANGE[2]), float(data_cfg.POINT_CLOUD_RANGE[5])
        c.w_fore_cls, c.w_mirr_cls = lw["occ_fore_cls_weight"], lw["occ_mirr_cls_weight"]
        c.w_bm_cls, c.w_neg_cls = lw["occ_bm_cls_weight"], lw["occ_neg_cls_weight"]
        c.w_fore_res, c.w_mirr_res = lw.get("occ_fore_res_weight", 0.1), lw.get("occ_mirr_res_weight", 0.1)
        c.w_bm_res, c.box_weight = lw.get("occ_bm_res_weight", 0.1), occ.BOX_WEIGHT
        c.reverse_vis = {"NOTHING": 0, "VCC": 1, "BACK_TRACK": 2}[self.reverse_vis]
        c.vis_half = (int(kern[2]) + 1) // 2
        self.fore_dropout_cls_weight = float(lw.get("fore_dropout_cls_weight", 0.0))
        self.fore_dropout_reg_weight = float(lw.get("fore_dropout_reg_weight", 0.0))
        self._cfg = c
        # back-projection of the occluded sphere cells (a10): "torch" (default) -- a static table made once with torch's own CPU
        # kernels, i.e. the arithmetic a CPU run of the reference quantises on (backproject_table_host: 0 cells differ from the
        # reference-pinned oracle); "device" -- the same table filled by the device's correctly-rounded transcendentals
        # (btc_occ_backproject_lut); "inline" -- no table, the kernel evaluates them per occluded cell (rounds 1-3).
        # OCC.TARGETS.BACKPROJECT in the model config.
        self.backproject = model_cfg.TARGETS.get("BACKPROJECT", "torch")
        assert self.backproject in ("torch", "device", "inline"), self.backproject
        self.sphere_offset = [float(v) for v in occ.get("SPHERE_OFFSET", [0.0, 0.0, 0.0])]
        # the reference adds the offset in BOTH directions (occ_pnts + sphere_offset going to the sphere grid, carte - offset coming back:
        # occ_targets_template.py:95,150); the device's forward projection (occupancy.hip occ_point_pass) does not take it, so a non-zero
        # offset would give targets that are silently inconsistent -- no shipped config sets one
        assert not any(self.sphere_offset), "a non-zero OCC.SPHERE_OFFSET is not implemented (the forward projection on the device ignores it)"
        self._lut = {}   # device -> table (plain attribute: not a buffer, so state_dict keys stay the reference's)

    def backproject_table(self, dev):
        """the static back-projection table on `dev` (None for BACKPROJECT: inline)"""
        if self.backproject == "inline":
            return None
        key = str(dev)
        lut = self._lut.get(key)
        if lut is None:
            c = self._cfg
            if self.backproject == "torch":
                lut = backproject_table_host(list(c.sphere_grid), list(c.sphere_range), list(c.sphere_voxel), list(c.grid), list(c.occ_range),
                                             list(c.occ_voxel), self.sphere_offset).to(dev)
            else:
                lut = torch.empty((self.sphere_nz, self.sphere_ny, self.sphere_nx), dtype=torch.int32, device=dev)
                check(lib().btc_occ_backproject_lut(ctypes.byref(c), ptr(lut), stream_ptr()), "btc_occ_backproject_lut")
            self._lut[key] = lut
        return lut

    def _dropout(self, batch_dict, out, coords, bs):
        """OCC.DROPOUT_RATE > 1e-3 in training (occ_targets_template.py:305-328, :342-343, :391-392): per scene a random share (uniform in
        [0, rate)) of the voxels is drawn WITH replacement; their payload is zeroed (DROPOUT_RMV: the voxels are removed) AFTER the targets
        were formed from all of them, and dropped foreground cells weigh more in both loss maps.  The draw uses numpy's and torch's
        generators as the reference does (`batch_dict['__dropped__']`, an (M,) bool tensor, replaces it: the tests compare on the
        reference's own draw).  An option path: a few torch ops and one read-back of the per-scene voxel counts."""
        dropped = batch_dict.pop("__dropped__", None)
        M = coords.shape[0]
        if dropped is None:
            ratios = np.random.uniform(high=self.dropout_rate, size=bs)
            counts = torch.bincount(coords[:, 0].long(), minlength=bs)[:bs].tolist()
            starts = np.concatenate([[0], np.cumsum(counts)])       # (scenes are contiguous in collate order; a general batch sorts first)
            order = torch.argsort(coords[:, 0].long(), stable=True)
            dropped = torch.zeros((M,), dtype=torch.bool, device=coords.device)
            for i in range(bs):
                k = int(counts[i] * ratios[i])
                if k > 0:
                    pick = torch.randint(low=0, high=int(counts[i]), size=[k])     # torch's CPU generator, as the reference's dropout() draws
                    dropped[order[int(starts[i]) + pick.to(coords.device)]] = True
        dropped = dropped.to(coords.device).bool()
        dc = coords[dropped].long()
        drop_mask = torch.zeros((bs, self.nz, self.ny, self.nx), dtype=torch.uint8, device=coords.device)
        drop_mask[dc[:, 0], dc[:, 1], dc[:, 2], dc[:, 3]] = 1
        fore_drop = out["fore_voxelwise_mask"] & drop_mask
        if self.fore_dropout_cls_weight > 1e-4:
            out["general_cls_loss_mask_float"] = out["general_cls_loss_mask_float"] + \
                (out["general_cls_loss_mask"] & fore_drop).to(torch.float32) * self.fore_dropout_cls_weight
        if self.fore_dropout_reg_weight > 1e-4:
            out["general_reg_loss_mask_float"] = out["general_reg_loss_mask_float"] + \
                (out["general_reg_loss_mask"] & fore_drop).to(torch.float32) * self.fore_dropout_reg_weight
        out["voxel_drop_mask"], out["fore_voxel_drop_mask"] = drop_mask, fore_drop
        if self.dropout_rmv:
            keep = ~dropped
            batch_dict['voxels'] = batch_dict['voxels'][keep]
            batch_dict['voxel_coords'] = batch_dict['voxel_coords'][keep]
            batch_dict['voxel_num_points'] = batch_dict['voxel_num_points'][keep]
            # (voxel_point_mask / final_point_mask keep their M rows, as in the reference: occ_targets_3d.py:30-40)
        else:
            batch_dict['voxels'][dropped] = 0

    def get_paddings_indicator(self, actual_num, max_num, axis=0):
        key = (int(max_num), actual_num.device)
        rng = self.__dict__.setdefault("_arange", {}).get(key)
        if rng is None:
            rng = self._arange[key] = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(1, -1)
        return actual_num.int().unsqueeze(1) > rng

    def forward(self, batch_dict, **kwargs):
        vox = batch_dict['voxels']
        dev = vox.device
        num = batch_dict['voxel_num_points'].int().contiguous()
        coords = batch_dict['voxel_coords'].int().contiguous()
        mask = self.get_paddings_indicator(num, vox.shape[1])
        batch_dict["voxel_point_mask"] = mask
        gt = batch_dict["gt_boxes"].float().contiguous()
        bs, G = gt.shape[0], gt.shape[1]
        gtn = batch_dict["gt_boxes_num"]
        gtn = gtn.to(dev).int() if torch.is_tensor(gtn) else torch.tensor(list(gtn), dtype=torch.int32, device=dev)
        mirr = batch_dict['box_mirr_flag'].float().contiguous()
        rot_z = batch_dict["rot_z"].float().contiguous() if "rot_z" in batch_dict else torch.zeros(bs, device=dev)
        bm = batch_dict.get("bm_points", None)
        n_bm = 0 if bm is None else int(bm.shape[0])
        bm = bm.float().contiguous() if n_bm > 0 else None
        # the kernels write the absolute coordinates into the voxel payload: a copy, unless the producer says nobody else holds the tensor
        # (DataProcessor.forward_batch: "__voxels_owned__")
        owned = bool(batch_dict.pop("__voxels_owned__", False)) and vox.is_contiguous() and vox.dtype == torch.float32
        if not owned:
            vox = vox.float().contiguous().clone() if not vox.is_contiguous() or vox.dtype != torch.float32 else vox.clone()
        M, P, C = vox.shape
        shape = (bs, self.nz, self.ny, self.nx)
        out = {}
        # ONE arena for what btc_occ_targets zeroes: [pos_all_num | the five byte masks the kernels accumulate into | the workspace] at the
        # offsets it recognises (csrc/occupancy.hip) -> one fill instead of three
        cfg = self._cfg
        cfg.batch, cfg.max_boxes = bs, G
        L = lib()
        ws_bytes = L.btc_occ_targets_ws_bytes(ctypes.byref(cfg))
        zeroed = OCC_BUFFER_FIELDS[:5]
        vol = bs * self.nz * self.ny * self.nx
        masks_bytes = (len(zeroed) * vol + 255) & ~255
        arena = torch.empty((256 + masks_bytes + max(int(ws_bytes), 256),), dtype=torch.uint8, device=dev)
        block = arena[256:256 + len(zeroed) * vol].view((len(zeroed),) + shape)
        ws = arena[256 + masks_bytes:]
        for i, k in enumerate(zeroed):
            out[k] = block[i]
        for k in OCC_BUFFER_FIELDS[5:]:
            if k == "res_mtrx":
                out[k] = torch.empty((bs, 3, self.nz, self.ny, self.nx), dtype=torch.float32, device=dev)
            elif k == "pos_all_num":
                out[k] = arena[0:4].view(torch.int32)
            elif k.endswith("_float"):
                out[k] = torch.empty(shape, dtype=torch.float32, device=dev)
            elif k == "forebox_label":
                out[k] = torch.empty(shape, dtype=torch.int8, device=dev)
            else:
                out[k] = torch.empty(shape, dtype=torch.uint8, device=dev)
        bufs = BtcOccBuffers(**{k: out[k].data_ptr() for k in OCC_BUFFER_FIELDS})
        lut = self.backproject_table(dev)
        cfg.backproject_lut = lut.data_ptr() if lut is not None else None
        check(L.btc_occ_targets(ctypes.byref(cfg), ptr(vox), ptr(coords), ptr(num), M, P, C, ptr(gt), ptr(gtn), ptr(mirr),
                                ptr(bm), n_bm, ptr(rot_z), ptr(self.all_voxel_centers), ctypes.byref(bufs), ptr(ws), ws_bytes,
                                stream_ptr()), "btc_occ_targets")
        batch_dict['voxels'] = vox  # absolute xyz payload (USE_ABSXYZ True)
        if self.dropout_rate > 1e-3 and batch_dict.get("is_train", True):
            self._dropout(batch_dict, out, coords, bs)
        out["occ_voxelwise_mask"] = out["occ_voxelwise_mask"].view(torch.bool)   # (0 / 1 bytes: the same memory as bool, no copy)
        out["pos_all_num"] = out["pos_all_num"][0]
        if not batch_dict.get("is_train", True):
            out["neg_mask"] = out["general_cls_loss_mask"] & (1 - out["pos_mask"])
        batch_dict.update(out)
        if "point_drop_inds" in batch_dict.keys():
            inds = batch_dict["point_drop_inds"]
            mask[inds[:, 0], inds[:, 1]] = False
        batch_dict["final_point_mask"] = mask
        return batch_dict


__all__ = {'OccTargets3D': OccTargets3D}
