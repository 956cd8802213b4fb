"""CPU ORACLE (test infrastructure only) for the occupancy / occlusion target generator, the VFEs,
PassOccVox and the occupancy losses -- a torch-CPU restatement of the reference's in-tree Python.

PARITY STATUS: pinned.  tests/test_oracle_golden.py checks every function here against
tests/golden/btc_small.npz, which was produced by running the REAL reference modules
(tests/golden/gen_golden.py).  Each function cites the reference lines it follows; floating-point
expressions keep the reference's operation order so the CPU results are reproduced exactly.
"""
import numpy as np
import torch
import torch.nn.functional as F

PI = np.pi

# Transcendentals.  Default = torch's own float32 CPU kernels, exactly what the reference runs on CPU (this is the
# mode pinned against the golden vectors).  Mode "cr" evaluates in float64 and rounds once (correctly rounded fp32):
# it is what btc_occ_targets computes on the GPU, and is used ONLY to separate kernel-logic errors from the
# last-ulp libm differences that the reference's boundary-aligned quantisation amplifies (tests/test_hip_occupancy.py).
_TRIG_CR = False


class trig_mode(object):
    def __init__(self, cr):
        self.cr = cr

    def __enter__(self):
        global _TRIG_CR
        self.prev, _TRIG_CR = _TRIG_CR, self.cr

    def __exit__(self, *a):
        global _TRIG_CR
        _TRIG_CR = self.prev


def _cos(v):
    return torch.cos(v.double()).to(v.dtype) if _TRIG_CR else torch.cos(v)


def _sin(v):
    return torch.sin(v.double()).to(v.dtype) if _TRIG_CR else torch.sin(v)


def _atan2(a, b):
    return torch.atan2(a.double(), b.double()).to(a.dtype) if _TRIG_CR else torch.atan2(a, b)



def cylinder_uvd2absxyz(u, v, d):
    """/root/reference/btcdet/utils/coords_utils.py:198-204"""
    return torch.stack([u * _cos(v * PI / 180.), -u * _sin(v * PI / 180.), d], dim=-1)


def sphere_uvd2absxyz(r, az, el):
    """coords_utils.py:180-186"""
    xyd = r * _cos(el * PI / 180.)
    return torch.stack([xyd * _cos(az * PI / 180.), -xyd * _sin(az * PI / 180.), r * _sin(el * PI / 180.)], dim=-1)


def cartesian_sphere_coords(p):
    """coords_utils.py:216-226 (perm xyz)"""
    sq = torch.square(p)
    dist = torch.sqrt(torch.sum(sq, dim=1))
    xyd = torch.sqrt(torch.sum(sq[..., 0:2], dim=-1))
    return torch.stack([dist, _atan2(-p[..., 1], p[..., 0]) * (180. / PI), _atan2(p[..., 2], xyd) * (180. / PI)], dim=-1)


def cartesian_cylinder_coords(p):
    """coords_utils.py:229-239 (perm xyz)"""
    sq = torch.square(p)
    xyd = torch.sqrt(torch.sum(sq[..., 0:2], dim=-1))
    return torch.stack([xyd, _atan2(-p[..., 1], p[..., 0]) * (180. / PI), p[..., 2]], dim=-1)


def yaw_rotation(yaw):
    """point_box_utils.py:310-319"""
    c, s, o, z = torch.cos(yaw), torch.sin(yaw), torch.ones_like(yaw), torch.zeros_like(yaw)
    return torch.stack([torch.stack([c, -1.0 * s, z], -1), torch.stack([s, c, z], -1), torch.stack([z, z, o], -1)], dim=-2)


def box_transform(rot, trans):
    """point_box_utils.py:323-329"""
    t = torch.cat([rot, trans.unsqueeze(-1)], dim=-1)
    last = torch.cat([torch.zeros_like(trans), torch.ones_like(trans[..., 0:1])], dim=-1)
    return torch.cat([t, last.unsqueeze(-2)], dim=-2)


def points_in_boxes(points, boxes):
    """point frame coords and in-box mask: point_box_utils.py:207-229 / 267-288.  -> pf (N,M,3), inbox (N,M) int8, R (M,3,3)"""
    center, dim, heading = boxes[:, :3], boxes[:, 3:6], boxes[:, 6]
    R = yaw_rotation(heading)
    T = torch.inverse(box_transform(R, center))
    pf = torch.einsum("nj,mij->nmi", points, T[:, :3, :3]) + T[:, :3, 3]
    inbox = torch.prod((pf <= dim * 0.5) & (pf >= -dim * 0.5), dim=-1, dtype=torch.int8)
    return pf, inbox, R


def rotatez(points, deg):
    """point_box_utils.py:241-249 (3-D and 2-D)"""
    a = deg * PI / 180.
    if points.shape[-1] == 3:
        rot = torch.transpose(yaw_rotation(a), 0, 1)
    else:
        c, s = torch.cos(a), torch.sin(a)
        rot = torch.transpose(torch.stack([torch.stack([c, -1.0 * s], -1), torch.stack([s, c], -1)], dim=-2), 0, 1)
    return torch.matmul(points, rot)


class OccOracle(object):
    """constants as OccTargetsTemplate.__init__ builds them (occ_targets_template.py:12-71)"""

    def __init__(self, cfg):
        d, m = cfg.DATA_CONFIG, cfg.MODEL.OCC
        self.cfg = cfg
        self.occ_range = np.asarray(d.OCC.POINT_CLOUD_RANGE, dtype=np.float32)
        self.det_range = d.POINT_CLOUD_RANGE
        vs = d.OCC.VOXEL_SIZE
        grid = np.round((self.occ_range[3:6] - self.occ_range[0:3]) / np.array(vs)).astype(np.int64)  # data_processor.py:119-120
        self.nx, self.ny, self.nz = [int(g) for g in grid]
        self.grid = grid
        self.vs = torch.as_tensor([vs], dtype=torch.float32)
        self.origin = torch.as_tensor([list(self.occ_range[:3])], dtype=torch.float32)
        self.pmax = torch.as_tensor([list(self.occ_range[3:])], dtype=torch.float32)
        self.max_grid = torch.as_tensor([[self.nx - 1, self.ny - 1, self.nz - 1]], dtype=torch.int64)
        self.min_grid = torch.zeros((1, 3), dtype=torch.int64)
        sr = np.asarray(d.OCC.SUPPORT_SPHERE_RANGE)
        svs = np.array([vs[0], vs[1], sr[6]])
        self.s_origin = torch.as_tensor(np.array([sr[:3]]), dtype=torch.float32)
        self.s_rev_origin = torch.as_tensor(np.array([sr[2::-1].copy()]), dtype=torch.float32)
        self.s_max = torch.as_tensor(np.array([sr[3:6]]), dtype=torch.float32)
        self.s_vs = torch.as_tensor(svs, dtype=torch.float32)
        self.s_rev_vs = torch.as_tensor([svs[2], svs[1], svs[0]], dtype=torch.float32)
        sg = ((sr[3:6] - sr[:3]) / svs).astype(int)
        self.snx, self.sny, self.snz = int(sg[0]), int(sg[1]), int(sg[2])
        self.s_max_grid = torch.as_tensor([[self.snx - 1, self.sny - 1, self.snz - 1]], dtype=torch.int64)
        kern = d.OCC.DIST_KERN
        self.kern = kern
        self.concede_x = d.OCC.get("CONCEDE_X", kern[-1] // 2 if d.OCC.get("HALF_X", False) else 0)
        lw = m.OCC_DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS
        self.lw = lw
        self.box_weight = d.OCC.BOX_WEIGHT
        self.empt_thresh = d.OCC.EMPT_SUR_THRESH
        self.reverse_vis = m.PARAMS.get("REVERSE_VIS", "NOTHING")        # occ_targets_template.py:66
        self.dropout_rate = d.OCC.get("DROPOUT_RATE", 0.0)
        self.dropout_rmv = d.OCC.get("DROPOUT_RMV", False)
        self.centers, self.centers_2d = self.voxel_centers()

    def voxel_centers(self):
        """detector3d_template.py:52-63 + coords_utils.py:166-177"""
        vs = torch.tensor([self.vs[0, 2], self.vs[0, 1], self.vs[0, 0]])
        org = torch.tensor([float(self.occ_range[2]), float(self.occ_range[1]), float(self.occ_range[0])])
        z, y, x = torch.meshgrid(torch.arange(self.nz), torch.arange(self.ny), torch.arange(self.nx), indexing="ij")
        zyx = torch.stack([z, y, x], dim=0)
        c = (0.5 + zyx.to(torch.float32)) * vs.view(3, 1, 1, 1) + org.view(3, 1, 1, 1)
        c = cylinder_uvd2absxyz(c[2], c[1], c[0])
        return c, torch.mean(c[:, :, :, :2], dim=0).view(-1, 2)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def point2coords_inrange(points, origin, pmax, max_grid, min_grid, vs):
        """occ_targets_template.py:82-90: inclusive range test, truncation, clamp"""
        ok = torch.cat([points[:, :3] >= origin, points[:, :3] <= pmax], dim=-1).all(-1)
        inds = torch.nonzero(ok)[..., 0]
        p = points[inds, :]
        c = ((p - origin) / vs).to(torch.int64)
        c = torch.maximum(torch.minimum(c, max_grid), min_grid)
        return c, inds

    def _mask(self, bs):
        return torch.zeros([bs, self.nz, self.ny, self.nx], dtype=torch.uint8)

    @staticmethod
    def _scatter1(mask, c):
        mask[c[..., 0], c[..., 1], c[..., 2], c[..., 3]] = 1
        return mask

    def vcc_mask(self, bs, coords):
        """occ_targets_template.py:432-447"""
        kz, ky, kx = self.kern
        z, y, x = torch.meshgrid(torch.arange(-(kz // 2), -(kz // 2) + kz), torch.arange(-(ky // 2), -(ky // 2) + ky),
                                 torch.arange(-(kx // 2) + self.concede_x, -(kx // 2) + self.concede_x + kx), indexing="ij")
        off = torch.stack([torch.zeros_like(z), z, y, x], dim=-1).view(1, -1, 4)
        c = (coords.view(-1, 1, 4) + off).view(-1, 4)
        m = self._mask(bs)
        m[c[:, 0].clamp(0, bs - 1), c[:, 1].clamp(0, self.nz - 1), c[:, 2].clamp(0, self.ny - 1), c[:, 3].clamp(0, self.nx - 1)] = 1
        return m

    def sphere_map(self, bs, pts, b, rot_z):
        """occ_targets_template.py:137-144 (+ the EMPT_SUR_THRESH fix :128-130,186-191) -> uint8 [B,49,157,214]"""
        sp = cartesian_sphere_coords(pts)
        sp[..., 1] += rot_z[b]
        smap = torch.zeros([bs, self.snz, self.sny, self.snx], dtype=torch.uint8)
        c, inds = self.point2coords_inrange(sp, self.s_origin, self.s_max, self.s_max_grid, self.min_grid, self.s_vs)
        bb = b[inds]
        smap[bb, c[..., 2], c[..., 1], c[..., 0]] = 1
        return smap

    def occluded_sphere(self, smap):
        """occ_from_sphere_ocp (occ_targets_template.py:110-134): which cells of the support sphere map count as occluded / unknown.
        NOTHING (configured): the EMPT_SUR_THRESH fix (in place on column 0, :128-130,186-191), then everything at or behind the first hit
        of a ray; VCC: everything except the DIST_KERN[2] + 1 >> 1 cells in front of a hit (unless hit themselves); BACK_TRACK: behind
        the LAST hit, or at / behind the first -- which differs from NOTHING only on rays without any hit (all of them count)."""
        if self.reverse_vis == "VCC":
            stride = self.kern[2] + 1
            inds = torch.nonzero(smap)
            inds = inds.unsqueeze(1).repeat(1, stride // 2, 1)
            inds[..., :, 3:4] -= torch.arange(1, stride // 2 + 1).view(1, stride // 2, 1).repeat(inds.shape[0], 1, 1)
            inds[..., :, 3] = torch.clamp(inds[..., :, 3], min=0, max=None)
            inds = inds.view(-1, 4)
            occ = torch.ones_like(smap)
            occ[inds[..., 0], inds[..., 1], inds[..., 2], inds[..., 3]] = 0
            return occ | smap
        if self.reverse_vis == "BACK_TRACK":
            rev = torch.flip(smap, [3])
            occ = torch.flip(torch.cumsum(rev, dim=3) < 0.9, [3])
            return occ | (torch.cumsum(smap, dim=3) > 0.9)
        if self.empt_thresh != "None" and self.empt_thresh < 9:
            cnt = torch.sum(smap, dim=3)
            nb = F.conv2d(cnt.unsqueeze(1).to(torch.float32), torch.ones(1, 1, 3, 3), padding=1) > self.empt_thresh
            smap[:, :, :, 0] = (cnt == 0) & nb.squeeze(1)
        return torch.cumsum(smap, dim=3) > 0.9

    def occ_from_cylin(self, bs, pts, b, rot_z):
        """occ_targets_template.py:136-155"""
        smap = self.sphere_map(bs, pts, b, rot_z)
        occl = self.occluded_sphere(smap)
        idx = torch.nonzero(occl)
        sb = idx[..., 0]
        sp = idx[..., 1:] * self.s_rev_vs + self.s_rev_origin
        xyz = sphere_uvd2absxyz(sp[..., 2], sp[..., 1], sp[..., 0])
        cyl = cartesian_cylinder_coords(xyz)
        c, inds = self.point2coords_inrange(cyl, self.origin, self.pmax, self.max_grid, self.min_grid, self.vs)
        m = self._mask(bs)
        m[sb[inds], c[..., 2], c[..., 1], c[..., 0]] = 1
        return m > 0.9, smap

    def filter_occ(self, occ, voxelwise):
        """occ_targets_template.py:249-255"""
        B, Z, Y, X = voxelwise.shape
        cz = self.centers[..., 2].unsqueeze(0)
        vz = (1 - voxelwise) * 100.0 + cz
        vz = torch.min(vz.view(B, Z * Y, X), dim=1, keepdim=True)[0].unsqueeze(1)
        vz -= (vz > 20.0) * 200
        return occ & (cz > torch.clamp(vz, min=self.det_range[2], max=None)) & (cz < self.det_range[5])

    def cell_center_xyz(self, coords, rot_z):
        """occ_targets_3d.py:133-145 (cylinder, rot=True)"""
        vc = (coords[:, [3, 2, 1]].float() + 0.5) * self.vs + self.origin
        vc[..., 1] -= rot_z[coords[:, 0]]
        return cylinder_uvd2absxyz(vc[..., 0], vc[..., 1], vc[..., 2])

    def mean_res(self, feat, coords, bs, rot_z):
        """occ_targets_3d.py:122-130"""
        out = torch.zeros([bs, 3, self.nz, self.ny, self.nx], dtype=torch.float32)
        if len(coords) > 0:
            uni, inv, cnt = torch.unique(coords, return_inverse=True, return_counts=True, dim=0)
            mean = torch.zeros([uni.shape[0], 3], dtype=feat.dtype).scatter_add_(0, inv.view(-1, 1).expand(-1, 3), feat[..., :3]) / cnt.float().unsqueeze(1)
            mean -= self.cell_center_xyz(uni, rot_z)
            out[uni[..., 0], :, uni[..., 1], uni[..., 2], uni[..., 3]] = mean
        return out

    def occ_coords(self, xyz, b, rot_z):
        """Cartesian points -> cylinder cell [b,z,y,x] of the (un-rotated) occupancy grid (occ_targets_3d.py:156-166)"""
        cyl = cartesian_cylinder_coords(xyz)
        cyl[..., 1] += rot_z[b]
        c, inds = self.point2coords_inrange(cyl, self.origin, self.pmax, self.max_grid, self.min_grid, self.vs)
        return torch.cat([b[inds].unsqueeze(-1), torch.stack([c[..., 2], c[..., 1], c[..., 0]], dim=-1)], dim=-1), inds

    # ------------------------------------------------------------------ OccTargets3D.forward (REG = True)
    def targets(self, bd, dropped=None):
        """occ_targets_3d.py:18-93 + occ_targets_template.py:330-401.  bd holds float32 tensors as after
        load_data_to_gpu (models/__init__.py:16-22); returns the batch_dict additions.
        dropped: (M,) bool -- the voxels OCC.DROPOUT_RATE > 1e-3 drops in training (occ_targets_template.py:305-328 draws them with
        np.random.uniform + torch.randint; the draw is an input here so that two implementations can be compared on the same one)"""
        vox, num, vcoords = bd['voxels'], bd['voxel_num_points'], bd['voxel_coords']
        gt, gtn, rot_z = bd["gt_boxes"], bd["gt_boxes_num"], bd["rot_z"]
        bs = gt.shape[0]
        out = {}
        mask = num.int().unsqueeze(1) > torch.arange(vox.shape[1], dtype=torch.int).view(1, -1)
        out["voxel_point_mask"] = mask
        occ_pnts = torch.cat([cylinder_uvd2absxyz(vox[..., 0], vox[..., 1], vox[..., 2]), vox[..., 3:]], dim=-1)
        out["voxels"] = occ_pnts
        gt = torch.cat([gt[..., :-1], (gt[..., -1:] > 1e-2).to(torch.float32)], dim=-1)
        vi = torch.nonzero(mask)
        vc = vcoords[vi[:, 0]].to(torch.int64)
        vf = occ_pnts[vi[:, 0], vi[:, 1]]
        voxelwise = self._scatter1(self._mask(bs), vc)
        vcc = self.vcc_mask(bs, vc)
        occ_raw, smap = self.occ_from_cylin(bs, vf[..., :3], vc[..., 0], rot_z)
        out["_sphere_map"] = smap
        out["_occ_raw"] = occ_raw
        occ = self.filter_occ(occ_raw, voxelwise)

        # ---- foreground / mirrored points (per scene, point_box_utils.py:70-97,252-306)
        label = torch.zeros(vf.shape[0], dtype=torch.int8)
        mpts, mb = [], []
        for i in range(bs):
            boxes, flag = gt[i, :gtn[i]], bd['box_mirr_flag'][i, :gtn[i]]
            sel = torch.nonzero(vc[:, 0] == i)[:, 0]
            if sel.numel() == 0:
                continue
            if boxes.shape[0] == 0:
                continue
            pf, inbox, R = points_in_boxes(vf[sel, :3], boxes)
            mi = torch.nonzero(inbox * (flag > 0.5).to(torch.int8).unsqueeze(0))
            mp = pf.clone()
            mp[:, :, 1] = -mp[:, :, 1]
            mp = torch.einsum("nmj,mij->nmi", mp, R) + boxes[:, :3]
            mpts.append(mp[mi[:, 0], mi[:, 1], :])
            mb.append(torch.full((mi.shape[0],), i, dtype=torch.int64))
            label[sel] = torch.max(inbox * boxes[..., 7].to(torch.int8).unsqueeze(0), dim=1)[0]
        fore = label > 0
        fore_c = vc[fore]
        fore_mask = self._scatter1(self._mask(bs), fore_c)
        fore_res = self.mean_res(vf[fore], fore_c, bs, rot_z)
        mirr_mask = self._mask(bs)
        mirr_res = torch.zeros([bs, 3, self.nz, self.ny, self.nx], dtype=torch.float32)
        if len(mpts) > 0:
            mp, mbi = torch.cat(mpts, 0), torch.cat(mb, 0)
            mc, inds = self.occ_coords(mp, mbi, rot_z)
            mirr_res = self.mean_res(mp[inds], mc, bs, rot_z)
            mirr_mask = self._scatter1(mirr_mask, mc)
        out["_mirr_mask_raw"] = mirr_mask.clone()
        mirr_mask = mirr_mask * (1 - voxelwise)
        mirr_res = mirr_res * (1 - voxelwise).unsqueeze(1)

        # ---- best-match template points (occ_targets_3d.py:96-119, point_box_utils.py:101-121)
        bm_mask = self._mask(bs)
        bm_res = torch.zeros([bs, 3, self.nz, self.ny, self.nx], dtype=torch.float32)
        bm = bd.get("bm_points", None)
        if bm is not None and len(bm) > 0:
            bb, bp = bm[..., 0].to(torch.int64), bm[..., 1:]
            lab = torch.zeros(bp.shape[0], dtype=torch.int8)
            for i in range(bs):
                sel = torch.nonzero(bb == i)[:, 0]
                boxes = gt[i, :gtn[i]]
                if sel.numel() > 0 and boxes.shape[0] > 0:
                    _, inbox, _ = points_in_boxes(bp[sel], boxes)
                    lab[sel] = torch.max(inbox * boxes[..., 7].to(torch.int8).unsqueeze(0), dim=1)[0]
            keep = torch.nonzero(lab)[..., 0]
            bb, bp = bb[keep], bp[keep, :]
            bc, inds = self.occ_coords(bp, bb, rot_z)
            bm_res = self.mean_res(bp[inds], bc, bs, rot_z)
            bm_mask = self._scatter1(bm_mask, bc)
        out["_bm_mask_raw"] = bm_mask.clone()
        bm_mask = bm_mask * (1 - voxelwise) * (1 - mirr_mask)
        bm_res = bm_res * (1 - voxelwise).unsqueeze(1) * (1 - mirr_mask).unsqueeze(1)

        # ---- cells inside GT boxes (occ_targets_3d.py:70-86)
        forebox = torch.zeros([bs, self.nz, self.ny, self.nx], dtype=torch.int8)
        for i in range(bs):
            boxes = gt[i, :gtn[i]]
            if boxes.shape[0] == 0:
                continue
            c2d = rotatez(self.centers_2d, rot_z[i])
            c2, dim2, R2 = boxes[:, :2], boxes[:, 3:5], boxes[:, 6]
            cs, sn = torch.cos(R2), torch.sin(R2)
            rot2 = torch.stack([torch.stack([cs, -1.0 * sn], -1), torch.stack([sn, cs], -1)], dim=-2)
            T2 = torch.inverse(box_transform(rot2, c2))
            pf2 = torch.einsum("nj,mij->nmi", c2d, T2[:, :2, :2]) + T2[:, :2, 2]
            in2 = (torch.prod((pf2 <= dim2 * 0.5) & (pf2 >= -dim2 * 0.5), dim=-1, dtype=torch.int8) > 0).any(-1)
            yx = in2.view(self.ny, self.nx).nonzero()
            if yx.shape[0] > 0:
                cen = rotatez(self.centers[:, yx[:, 0], yx[:, 1], ...].reshape(-1, 3), rot_z[i])
                _, inbox, _ = points_in_boxes(cen, boxes)
                lab = torch.max(inbox * boxes[..., 7].to(torch.int8).unsqueeze(0), dim=1)[0]
                forebox[i, :, yx[:, 0], yx[:, 1]] = lab.view(self.nz, -1)

        # ---- loss maps (occ_targets_template.py:330-401)
        lw = self.lw
        gen = vcc & occ
        f_cls, m_cls, b_cls = fore_mask & gen, mirr_mask & gen, bm_mask & gen
        pos = f_cls | m_cls | b_cls
        neg = gen & (1 - pos)
        cls_w = f_cls.float() * lw["occ_fore_cls_weight"] + m_cls.float() * lw["occ_mirr_cls_weight"] + \
            b_cls.float() * lw["occ_bm_cls_weight"] + neg.float() * lw["occ_neg_cls_weight"]
        if self.box_weight != 1.0:
            cls_w += (neg & (forebox > 1e-3)).float() * (self.box_weight - lw["occ_neg_cls_weight"])
        reg_w = f_cls.float() * lw.get("occ_fore_res_weight", 0.1) + m_cls.float() * lw.get("occ_mirr_res_weight", 0.1) + \
            b_cls.float() * lw.get("occ_bm_res_weight", 0.1)
        reg_m = (reg_w > 0).to(torch.uint8)
        if dropped is not None and self.dropout_rate > 1e-3 and bd.get("is_train", True):
            # dropout (occ_targets_template.py:305-328): the dropped voxels' payload is zeroed (or the voxels removed, DROPOUT_RMV) AFTER the
            # targets were formed from all of them; dropped foreground cells weigh more in both losses (:342-343, :391-392)
            dc = vcoords[dropped].to(torch.int64)
            drop_mask = self._scatter1(self._mask(bs), dc)
            fore_drop = fore_mask & drop_mask
            if lw["fore_dropout_cls_weight"] > 1e-4:
                cls_w = cls_w + (gen & fore_drop).float() * lw["fore_dropout_cls_weight"]
            if lw["fore_dropout_reg_weight"] > 1e-4:
                reg_w = reg_w + (reg_m & fore_drop).float() * lw["fore_dropout_reg_weight"]
            out["voxel_drop_mask"], out["fore_voxel_drop_mask"] = drop_mask, fore_drop
            if self.dropout_rmv:
                keep = ~dropped
                out["voxels"], out["_voxel_coords"], out["_voxel_num_points"] = out["voxels"][keep], vcoords[keep], num[keep]
            else:
                out["voxels"] = out["voxels"].clone()
                out["voxels"][dropped] = 0
        res = fore_res * reg_m.unsqueeze(1) + mirr_res * reg_m.unsqueeze(1) + bm_res * reg_m.unsqueeze(1)
        out.update({"vcc_mask": vcc, "voxelwise_mask": voxelwise, "bm_voxelwise_mask": bm_mask, "occ_voxelwise_mask": occ,
                    "fore_voxelwise_mask": fore_mask, "pos_mask": pos, "general_cls_loss_mask": gen,
                    "pos_all_num": torch.sum(fore_mask | mirr_mask | bm_mask), "occ_fore_cls_mask": f_cls,
                    "occ_mirr_cls_mask": m_cls, "occ_bm_cls_mask": b_cls, "forebox_label": forebox,
                    "general_cls_loss_mask_float": cls_w, "general_reg_loss_mask": reg_m, "general_reg_loss_mask_float": reg_w,
                    "res_mtrx": res, "final_point_mask": mask,
                    "_fore_res": fore_res, "_mirr_res": mirr_res, "_bm_res": bm_res, "_mirr_mask": mirr_mask})
        return out


# ---------------------------------------------------------------------- VFEs (mean_vfe.py:38-44, occ_vfe.py:36-53)
def mean_vfe(voxels, num):
    return (voxels.sum(dim=1) / torch.clamp_min(num.view(-1, 1), min=1.0).type_as(voxels)).contiguous()


def occ_vfe(voxels, num, n_raw=4):
    mask = torch.arange(voxels.shape[1], dtype=torch.int).view(1, -1) < num.int().view(-1, 1)
    raw = (voxels[:, :, -1] < 0.05) & mask
    occ = (voxels[:, :, -1] >= 0.05) & mask
    rn, on = raw.sum(dim=1).view(-1, 1), occ.sum(dim=1).view(-1, 1)
    occ_only = (on > 0.5) & (rn < 0.5)
    rf = (raw.unsqueeze(-1) * voxels[:, :, :n_raw]).sum(dim=1) / torch.clamp_min(rn, min=1.0).type_as(voxels)
    of = (occ.unsqueeze(-1) * voxels[:, :, :n_raw]).sum(dim=1) / torch.clamp_min(on, min=1.0).type_as(voxels)
    omax = voxels[:, :, n_raw:].max(dim=1)[0]
    return torch.cat([rf + occ_only * of, omax], dim=-1), omax


# ---------------------------------------------------------------------- occupancy losses (occ_head_template.py:52-111)
def occ_losses(bd, lw):
    def masked(pred, target, fn, w, mask):
        i = mask.nonzero()
        w = w[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]]
        loss = fn(pred[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]], target[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]]) * w
        return torch.sum(loss) / torch.clamp(torch.sum(w), min=1.0)

    def focal(x, t):  # loss_utils.py:140-152 with eps = 1e-6 (:169), alpha 1, gamma 2
        p = F.softmax(x, dim=1) + 1e-6
        return torch.sum(t * (-1.0 * torch.pow(-p + 1., 2.0) * torch.log(p)), dim=1, keepdim=True)

    def smooth_l1(x, t, beta=lw['res_beta']):  # loss_utils.py:199-233
        n = torch.abs(x - t)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    pos = bd["pos_mask"].to(torch.float32)
    onehot = torch.stack([1.0 - pos, pos], dim=-1).permute(0, 4, 1, 2, 3)
    cls = masked(bd['pred_occ_logit'], onehot, focal, bd["general_cls_loss_mask_float"].unsqueeze(1),
                 bd['general_cls_loss_mask']) * lw.get("occ_fore_cls_weight", 1.0)
    reg = masked(bd['pred_sem_residuals'], bd['res_mtrx'], smooth_l1, bd["general_reg_loss_mask_float"].unsqueeze(1),
                 bd["general_reg_loss_mask"]) * lw.get("occ_fore_res_weight", 0.1)
    return cls + reg, cls, reg


# ---------------------------------------------------------------------- PassOccVox (pass_occ_vox.py:10-59)
def pass_occ_vox(bd, cfg, oracle_revoxelize):
    """returns voxels, num, coords, occ_pnts, added_occ_b_ind.  Cells above OCC_THRESH (top-k 2048 per scene when
    more) -> centre (+ residual) -> xyz -> detection cell; merged with the detection voxels' points.
    The reference's topk(sorted=False) order is unspecified; selected cells are kept in topk's returned order here
    (as the reference does on CPU) so that the golden vectors reproduce."""
    d, p = cfg.DATA_CONFIG, cfg.MODEL.OCC.PARAMS
    occ_range, vs = d.OCC.POINT_CLOUD_RANGE, d.OCC.VOXEL_SIZE
    det_range = torch.tensor(d.POINT_CLOUD_RANGE, dtype=torch.float32)
    det_vs = torch.tensor(d.DATA_PROCESSOR[3].VOXEL_SIZE, dtype=torch.float32)
    det_grid = np.round((np.array(d.POINT_CLOUD_RANGE[3:6], np.float32) - np.array(d.POINT_CLOUD_RANGE[0:3], np.float32))
                        / np.array(d.DATA_PROCESSOR[3].VOXEL_SIZE)).astype(np.int64)
    probs, res = bd['batch_pred_occ_prob'], bd["pred_sem_residuals"]
    pl, rl, cl = [], [], []
    for i in range(bd['batch_size']):
        m = probs[i] > p.OCC_THRESH
        c = torch.nonzero(m)
        if int(m.sum()) > 0 and bd["use_occ_prob"][i]:
            tp, tr = probs[i][m], res[i][:, m].permute(1, 0)
            if tp.shape[0] > p.MAX_NUM_OCC_PNTS:
                tp, ti = torch.topk(tp, p.MAX_NUM_OCC_PNTS, largest=True, sorted=False)
                c, tr = c[ti, ...], tr[ti, ...]
            cl.append(torch.cat([torch.full_like(c[..., :1], i), c], dim=-1))
            pl.append(tp)
            rl.append(tr)
    oc, op, orr = torch.cat(cl, 0), torch.cat(pl, 0), torch.cat(rl, 0)
    cx = occ_range[0] + (oc[..., 3] + 0.5) * vs[0]          # add_occ_template.py:131-146
    cy = occ_range[1] + (oc[..., 2] + 0.5) * vs[1]
    cz = occ_range[2] + (oc[..., 1] + 0.5) * vs[2]
    cy = cy - bd["rot_z"][oc[..., 0]]
    xyz = cylinder_uvd2absxyz(cx, cy, cz) + orr
    occ_pnts = torch.cat([xyz, op.unsqueeze(-1)], dim=-1)
    cc = torch.div(xyz - det_range[0:3].unsqueeze(0), det_vs.unsqueeze(0))   # add_occ_template.py:78-88
    gx = torch.clamp(torch.floor(cc[..., 0]), min=0, max=int(det_grid[0]) - 1).to(torch.int64)
    gy = torch.clamp(torch.floor(cc[..., 1]), min=0, max=int(det_grid[1]) - 1).to(torch.int64)
    gz = torch.clamp(torch.floor(cc[..., 2]), min=0, max=int(det_grid[2]) - 1).to(torch.int64)
    occ_cells = torch.stack([oc[..., 0], gz, gy, gx], dim=-1)
    occ6 = torch.cat([xyz, torch.zeros_like(op).unsqueeze(-1), op.unsqueeze(-1), torch.ones_like(op).unsqueeze(-1)], dim=-1)
    dv, dn, dc = bd['det_voxels'], bd['det_voxel_num_points'], bd['det_voxel_coords']
    mask = dn.unsqueeze(1) > torch.arange(dv.shape[1], dtype=torch.int).view(1, -1)   # add_occ_template.py:168-190
    ii = mask.nonzero()
    gp = torch.cat([dv[ii[:, 0], ii[:, 1], :], torch.zeros(ii.shape[0], 2)], dim=-1)
    gc = dc[ii[:, 0], :].to(torch.int64)
    v, n, c = oracle_revoxelize(torch.cat([gp, occ6], 0).numpy(), torch.cat([gc, occ_cells], 0).numpy())
    return v, n, c, occ_pnts.numpy(), oc[..., 0].numpy()
