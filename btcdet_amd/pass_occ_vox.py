"""PassOccVox: occupancy probabilities -> added occupancy points -> merged detection voxels.

Mirror of /root/reference/btcdet/models/occ_pnt/pass_occ_vox.py:10-59 and
add_occ_template.py:78-190,248-268 (module protocol ``forward(batch_dict) -> batch_dict``).
The GPU re-voxelization (torch.unique(dim=0) + sort + scatter-pad + .cpu() sync in the reference)
is one HIP call pair (btc_revoxelize_count / btc_revoxelize_fill).
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, i3, i3p, lib, ptr, stream_ptr, workspace


FUSED = True  # PassOccVox.forward through btc_pass_occ_vox_* (False: the torch op chain + btc_revoxelize_*)


def revoxelize(points, coords, batch_size, grid_zyx):
    """combine_gt_occ_voxel_point (add_occ_template.py:262-268) on the GPU.
    points (n,C) f32, coords (n,4) int64 [b,z,y,x] -> voxels (M,Pmax,C) f32, num (M,) i64, vcoords (M,4) i64
    (cells ascending in (b,z,y,x), points of a cell in input order, zero padded)."""
    points = points.contiguous()
    coords = coords.contiguous()
    if coords.dtype != torch.int64 or points.dtype != torch.float32:
        raise _lib.BtcHipError("revoxelize: int64 coords and float32 points expected")
    n, C = points.shape
    dev = points.device
    sh = i3([int(v) for v in grid_zyx])
    L = lib()
    ws_bytes = L.btc_revoxelize_ws_bytes(n, int(batch_size), i3p(sh))
    ws = workspace(ws_bytes, dev)
    d_mp = torch.zeros((2,), dtype=torch.int32, device=dev)
    check(L.btc_revoxelize_count(ptr(coords), n, int(batch_size), i3p(sh), ptr(d_mp[0:1]), ptr(d_mp[1:2]), ptr(ws),
                                 ws_bytes, stream_ptr()), "btc_revoxelize_count")
    m, pmax = [int(v) for v in d_mp.tolist()]  # the one read-back (the reference syncs on Pmax too, :251)
    voxels = torch.empty((m, pmax, C), dtype=torch.float32, device=dev)
    vcoords = torch.empty((m, 4), dtype=torch.int64, device=dev)
    vnum = torch.empty((m,), dtype=torch.int64, device=dev)
    check(L.btc_revoxelize_fill(ptr(points), ptr(coords), n, C, int(batch_size), i3p(sh), m, pmax, ptr(voxels),
                                ptr(vcoords), ptr(vnum), ptr(ws), ws_bytes, stream_ptr()), "btc_revoxelize_fill")
    return voxels, vnum, vcoords


class PassOccVox(torch.nn.Module):
    """Module protocol of the reference (constructor kwargs of detector3d_template.py:133-154)."""

    def __init__(self, model_cfg, data_cfg, point_cloud_range, occ_voxel_size, occ_grid_size, det_voxel_size, det_grid_size,
                 mode, voxel_centers, **kwargs):
        super().__init__()
        self.model_cfg, self.data_cfg = model_cfg, data_cfg
        p = model_cfg.PARAMS
        self.occ_thresh, self.eval_occ_thresh = p.OCC_THRESH, p.EVAL_OCC_THRESH
        self.max_add_occpnts_num, self.eval_max_add_occpnts_num = p.MAX_NUM_OCC_PNTS, p.EVAL_MAX_NUM_OCC_PNTS
        self.pass_gradient = model_cfg.OCC_PNT_UPDATE.PASS_GRAD
        self.res_num_dim = data_cfg.OCC.RES_NUM_DIM
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.occ_voxel_size = occ_voxel_size
        self.nvx, self.nvy, self.nvz = [float(v) for v in occ_voxel_size]
        self.occ_grid_size, self.det_grid_size = occ_grid_size, det_grid_size
        self.det_voxel_size = [float(v) for v in det_voxel_size]
        self.occ_point_cloud_range = data_cfg.OCC.POINT_CLOUD_RANGE
        self.occ_x_origin, self.occ_y_origin, self.occ_z_origin = [float(v) for v in self.occ_point_cloud_range[:3]]
        self.all_voxel_centers = voxel_centers["all_voxel_centers"]
        self.config_realdrop = data_cfg.OCC.get('REAL_DROP', None) is None or data_cfg.OCC.REAL_DROP
        self.config_rawadd = data_cfg.OCC.get('RAW_ADD', False)
        self.code_num_dim = data_cfg.OCC.get('CODE_NUM_DIM', 2)
        self.reg = p.get("REG", False)
        self.db_proj = model_cfg.OCC_PNT_UPDATE.get('DB_PROJ', False)
        self.remain_percentage = p.get('REMAIN_PERCENTAGE', None)
        assert not self.db_proj and self.remain_percentage is None, "DB_PROJ / REMAIN_PERCENTAGE are off in the configured model"
        assert data_cfg.OCC.COORD_TYPE == "cylinder"

    def visualize(self, batch_dict, binds):  # visualisation is out of scope
        return {}, {}

    def filter_occ_points(self, batch_size, occ_probs, batch_dict):
        """cells with p > OCC_THRESH, at most MAX_NUM_OCC_PNTS per scene (highest p); add_occ_template.py:94-128.
        (The reference compares against self.occ_thresh in both modes, App. D.2.)  The selected cells of a scene are
        kept in ascending cell order -- topk(sorted=False) leaves the order unspecified in the reference."""
        max_add = self.max_add_occpnts_num if batch_dict["is_train"] else self.eval_max_add_occpnts_num
        res_lst, probs_lst, coords_lst = [], [], []
        for i in range(batch_size):
            if not batch_dict["use_occ_prob"][i]:
                continue
            flat = occ_probs[i].reshape(-1)
            sel = torch.nonzero(flat > self.occ_thresh)[:, 0]
            if sel.numel() == 0:
                continue
            if sel.numel() > max_add:
                top = torch.topk(flat[sel], max_add, largest=True, sorted=False)[1]
                sel = sel[torch.sort(top)[0]]
            NZ, NY, NX = occ_probs.shape[1:]
            z, r = torch.div(sel, NY * NX, rounding_mode='floor'), sel % (NY * NX)
            coords_lst.append(torch.stack([torch.full_like(sel, i), z, torch.div(r, NX, rounding_mode='floor'), r % NX], dim=-1))
            probs_lst.append(flat[sel])
            if self.reg:
                res_lst.append(batch_dict["pred_sem_residuals"][i].reshape(self.res_num_dim, -1)[:, sel].permute(1, 0))
        return res_lst, probs_lst, coords_lst

    def occ_coords2absxyz(self, occ_coords, type, rot_z=None):
        """cell centre of the cylinder grid -> Cartesian (add_occ_template.py:131-146)"""
        cx = self.occ_x_origin + (occ_coords[..., 3] + 0.5) * self.nvx
        cy = self.occ_y_origin + (occ_coords[..., 2] + 0.5) * self.nvy
        cz = self.occ_z_origin + (occ_coords[..., 1] + 0.5) * self.nvz
        if rot_z is not None:
            cy = cy - rot_z[occ_coords[..., 0]]
        return torch.stack([cx * torch.cos(cy * np.pi / 180.), -cx * torch.sin(cy * np.pi / 180.), cz], dim=-1)

    def trans_voxel_grid(self, occ_xyz, b_inds):
        """xyz -> detection-grid cell [b,z,y,x], floor + clamp (add_occ_template.py:78-88)"""
        rng = torch.tensor(self.point_cloud_range[0:3], device=occ_xyz.device, dtype=torch.float32)
        vs = torch.tensor(self.det_voxel_size, device=occ_xyz.device, dtype=torch.float32)
        c = torch.floor(torch.div(occ_xyz - rng.unsqueeze(0), vs.unsqueeze(0)))
        nx, ny, nz = [int(g) for g in self.det_grid_size]
        cx = torch.clamp(c[..., 0], min=0, max=nx - 1).to(torch.int64)
        cy = torch.clamp(c[..., 1], min=0, max=ny - 1).to(torch.int64)
        cz = torch.clamp(c[..., 2], min=0, max=nz - 1).to(torch.int64)
        return torch.stack([b_inds, cz, cy, cx], dim=-1)

    def _fused_ok(self, batch_dict):
        fpm = batch_dict.get("final_point_mask", None)
        dv = batch_dict.get('det_voxels', None)
        realdrop = self.config_realdrop and fpm is not None and dv is not None and dv.shape[1] == fpm.shape[1]
        return FUSED and 'det_voxel_coords' in batch_dict and dv is not None and dv.is_cuda and not realdrop \
            and self.res_num_dim == 3 and dv.shape[2] >= 4

    def _pov_config(self, bs, is_train):
        """BtcPovConfig for (batch size, mode), built once"""
        from ._lib import BtcPovConfig
        memo = self.__dict__.setdefault("_pov_cfg_memo", {})
        c = memo.get((bs, is_train))
        if c is None:
            c = BtcPovConfig()
            c.batch = bs
            c.max_k = int(self.max_add_occpnts_num if is_train else self.eval_max_add_occpnts_num)
            c.occ_grid[:] = [int(g) for g in self.occ_grid_size]
            c.det_grid[:] = [int(g) for g in self.det_grid_size]
            c.occ_origin[:] = [self.occ_x_origin, self.occ_y_origin, self.occ_z_origin]
            c.occ_voxel[:] = [self.nvx, self.nvy, self.nvz]
            c.det_origin[:] = self.point_cloud_range[0:3]
            c.det_voxel[:] = self.det_voxel_size
            c.occ_thresh = float(self.occ_thresh)
            c.inten = float(self.data_cfg.OCC.INTEN if self.data_cfg.OCC.get("INTEN", None) is not None else 0.0)
            c.code_dim = int(self.code_num_dim)
            memo[(bs, is_train)] = c
        return c

    def forward_fused(self, batch_dict):
        """the whole module as two C-ABI calls around one read-back (csrc/pass_occ.hip)"""
        import ctypes
        bs, probs = batch_dict['batch_size'], batch_dict['batch_pred_occ_prob'].contiguous()
        dv = batch_dict['det_voxels'].float().contiguous()
        dn = batch_dict['det_voxel_num_points'].int().contiguous()
        dc = batch_dict['det_voxel_coords'].int().contiguous()
        dev = dv.device
        M, P, C = dv.shape
        res = batch_dict["pred_sem_residuals"].detach().contiguous() if self.reg else None
        is_train = batch_dict["is_train"]
        c = self._pov_config(bs, bool(is_train))
        use_np = np.asarray(batch_dict["use_occ_prob"], dtype=bool)
        # every scene takes occupancy points (USEOCC_PERCENTAGE >= 1): NULL = all set, no host-to-device copy
        use = None if bool(use_np.all()) else torch.as_tensor(use_np.astype(np.uint8)).to(dev)
        rot = batch_dict["rot_z"].float().contiguous() if "rot_z" in batch_dict else None
        L = lib()
        ws_bytes = L.btc_pass_occ_vox_ws_bytes(ctypes.byref(c), M, P)
        ws = workspace(ws_bytes, dev)
        d_info = torch.empty((2 + bs,), dtype=torch.int32, device=dev)
        check(L.btc_pass_occ_vox_count(ctypes.byref(c), ptr(probs.detach()), ptr(res), ptr(use), ptr(rot), ptr(dc), ptr(dn), M, P, C,
                                       ptr(d_info), ptr(ws), ws_bytes, stream_ptr()), "btc_pass_occ_vox_count")
        info = d_info.tolist()  # the one read-back of the module
        m, pmax, k_total = info[0], info[1], sum(info[2:])
        batch_dict["gt_points_xyz"] = batch_dict["points"][..., 1:4]
        batch_dict["gt_b_ind"] = batch_dict["points"][..., 0]
        if k_total == 0:  # nothing passed the threshold: detection voxels + zero code channels (pass_occ_vox.py:48-53)
            batch_dict['voxel_num_points'], batch_dict['voxel_coords'] = batch_dict['det_voxel_num_points'], batch_dict['det_voxel_coords']
            batch_dict["added_occ_b_ind"] = torch.zeros([1], dtype=torch.int64, device=dev)
            batch_dict["added_occ_xyz"] = torch.zeros([1, 3], dtype=torch.float32, device=dev)
            batch_dict["occ_pnts"] = torch.zeros([1, 4], dtype=torch.float32, device=dev)
            batch_dict['voxels'] = torch.cat((dv, torch.zeros_like(dv[..., 0:self.code_num_dim])), dim=-1)
            return batch_dict
        voxels = torch.empty((m, pmax, C + self.code_num_dim), dtype=torch.float32, device=dev)
        vcoords = torch.empty((m, 4), dtype=torch.int64, device=dev)
        vnum = torch.empty((m,), dtype=torch.int64, device=dev)
        occ_pnts = torch.empty((k_total, 4), dtype=torch.float32, device=dev)
        occ_b = torch.empty((k_total,), dtype=torch.int64, device=dev)
        # int32 twins of the coordinates / counts for the modules behind this one (OccVFE, the detection backbone): the int64 tensors are
        # what the reference's torch.unique hands on, and converting them cost a launch each
        vcoords32 = torch.empty((m, 4), dtype=torch.int32, device=dev)
        vnum32 = torch.empty((m,), dtype=torch.int32, device=dev)
        check(L.btc_pass_occ_vox_fill_i32(ctypes.byref(c), ptr(dv), M, P, C, m, pmax, k_total, ptr(voxels), ptr(vcoords), ptr(vnum),
                                          ptr(occ_pnts), ptr(occ_b), ptr(vcoords32), ptr(vnum32), ptr(ws), ws_bytes, stream_ptr()),
              "btc_pass_occ_vox_fill_i32")
        batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords'] = voxels, vnum, vcoords
        batch_dict['__voxel_i32__'] = (vcoords, vcoords32, vnum, vnum32)   # (tensor, its int32 twin) pairs, matched by identity
        batch_dict["occ_pnts"], batch_dict["added_occ_xyz"], batch_dict["added_occ_b_ind"] = occ_pnts, occ_pnts[:, :3], occ_b
        return batch_dict

    def forward(self, batch_dict, **kwargs):
        if self._fused_ok(batch_dict) and not self.pass_gradient:
            return self.forward_fused(batch_dict)
        pnt_feat_dim = batch_dict['voxels'].shape[2]
        batch_size, probs = batch_dict['batch_size'], batch_dict['batch_pred_occ_prob']
        res_lst, probs_lst, coords_lst = self.filter_occ_points(batch_size, probs, batch_dict)
        batch_dict["gt_points_xyz"] = batch_dict["points"][..., 1:4]
        batch_dict["gt_b_ind"] = batch_dict["points"][..., 0]
        if 'det_voxel_coords' in batch_dict:
            batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords'] = \
                batch_dict['det_voxels'], batch_dict['det_voxel_num_points'], batch_dict['det_voxel_coords']
        dev = batch_dict['voxels'].device
        if len(probs_lst) > 0:
            occ_probs, occ_coords = torch.cat(probs_lst, dim=0), torch.cat(coords_lst, dim=0)
            xyz = self.occ_coords2absxyz(occ_coords, self.data_cfg.OCC.COORD_TYPE, rot_z=batch_dict.get("rot_z", None))
            if self.reg:
                xyz = xyz + torch.cat(res_lst, dim=0)
            batch_dict["added_occ_xyz"] = xyz
            batch_dict["occ_pnts"] = torch.cat([xyz, occ_probs.unsqueeze(-1)], dim=-1)
            batch_dict["added_occ_b_ind"] = occ_coords[..., 0]
            occ_cells = self.trans_voxel_grid(xyz, occ_coords[..., 0])
            # [x, y, z, intensity = OCC.INTEN, prob, 1] (assemble_occ_points, add_occ_template.py:149-165)
            inten = self.data_cfg.OCC.INTEN if self.data_cfg.OCC.get("INTEN", None) is not None else 0.0
            cols = [xyz]
            if self.res_num_dim < pnt_feat_dim:
                cols.append(torch.full_like(occ_probs, inten).unsqueeze(-1))
                if pnt_feat_dim > 4:
                    cols.append(torch.zeros_like(occ_probs).unsqueeze(-1))
            cols.append(occ_probs.unsqueeze(-1))
            if self.code_num_dim > 1:
                cols.append(torch.ones_like(occ_probs).unsqueeze(-1))
            occ_points = torch.cat(cols, dim=-1)
            # valid points of the detection voxels + zero code channels (assemble_gt_vox_points, :168-190)
            gv, gn, gc = batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords']
            fpm = batch_dict.get("final_point_mask", None)
            if self.config_realdrop and fpm is not None and gv.shape[1] == fpm.shape[1]:
                mask = fpm
            else:
                mask = gn.view(-1, 1) > torch.arange(gv.shape[1], dtype=torch.int, device=dev).view(1, -1)
            inds = mask.nonzero()
            gt_points = torch.cat([gv[inds[:, 0], inds[:, 1], :],
                                   torch.zeros((inds.shape[0], self.code_num_dim), dtype=gv.dtype, device=dev)], dim=-1)
            gt_cells = gc[inds[:, 0], :].to(torch.int64)
            points = torch.cat((gt_points, occ_points), dim=0)
            cells = torch.cat((gt_cells, occ_cells), dim=0)
            nx, ny, nz = [int(g) for g in self.det_grid_size]
            voxels, num, vcoords = revoxelize(points.float(), cells, batch_size, [nz, ny, nx])
            batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords'] = voxels, num, vcoords
        else:
            zeros = torch.zeros_like(batch_dict['voxels'][..., 0:self.code_num_dim])
            batch_dict["added_occ_b_ind"] = torch.zeros([1], dtype=torch.int64, device=dev)
            batch_dict["added_occ_xyz"] = torch.zeros([1, 3], dtype=torch.float32, device=dev)
            batch_dict["occ_pnts"] = torch.zeros([1, 4], dtype=torch.float32, device=dev)
            batch_dict['voxels'] = torch.cat((batch_dict['voxels'], zeros), dim=-1)
        if not self.pass_gradient:
            for k in ('occ_pnts', 'added_occ_xyz', 'added_occ_b_ind', 'voxels'):
                batch_dict[k] = batch_dict[k].detach()
        return batch_dict


__all__ = {'PassOccVox': PassOccVox}
