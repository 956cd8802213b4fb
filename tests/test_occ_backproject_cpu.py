"""The static back-projection table of the occlusion targets (btcdet_amd/occ_targets.py backproject_table_host, a10) against the
reference-pinned oracle: for every golden batch the occlusion mask derived from the oracle's sphere map THROUGH THE TABLE equals
the oracle's own back-projection (occ_targets_template.py:146-154, evaluated per batch on the occluded cells only) bit for bit.
The oracle itself is pinned to the reference's run by tests/test_oracle_golden.py / test_golden_full_cpu.py."""
import numpy as np
import pytest
import torch

from golden_batch import golden_batch, golden_batch_full
from oracle import occ_oracle

from btcdet_amd.config import load_cfg
from btcdet_amd.occ_targets import backproject_table_host


def _table(O):
    d = O.cfg.DATA_CONFIG
    sr = np.asarray(d.OCC.SUPPORT_SPHERE_RANGE)
    vs = d.OCC.VOXEL_SIZE
    return backproject_table_host([O.snx, O.sny, O.snz], [float(np.float32(v)) for v in sr[:6]],
                                  [float(np.float32(v)) for v in (vs[0], vs[1], sr[6])], [O.nx, O.ny, O.nz],
                                  [float(np.float32(v)) for v in O.occ_range], [float(np.float32(v)) for v in vs])


@pytest.mark.parametrize("which", ["golden", "full_a", "full_b", "full_c", "full_d"])
def test_table_reproduces_the_reference_back_projection(which):
    cfg = load_cfg()
    O = occ_oracle.OccOracle(cfg)
    lut = _table(O).view(-1).long()
    assert lut.shape[0] == O.snz * O.sny * O.snx and int(lut.max()) < O.nz * O.ny * O.nx and int(lut.min()) == -1
    bd = golden_batch()[2] if which == "golden" else golden_batch_full(which)[2]
    ref = O.targets(bd)
    smap = ref["_sphere_map"]
    bs = smap.shape[0]
    occl = torch.cumsum(smap, dim=3) > 0.9
    m = torch.zeros(bs, O.nz * O.ny * O.nx, dtype=torch.bool)
    for b in range(bs):
        cells = lut[occl[b].view(-1)]
        m[b, cells[cells >= 0]] = True
    assert torch.equal(m.view(bs, O.nz, O.ny, O.nx), ref["_occ_raw"].bool())


def test_table_is_not_the_correctly_rounded_one():
    """what the table is for: with correctly-rounded transcendentals thousands of boundary corners land in the neighbouring cell"""
    O = occ_oracle.OccOracle(load_cfg())
    a = _table(O)
    with occ_oracle.trig_mode(True):
        z, y, x = torch.meshgrid(torch.arange(O.snz), torch.arange(O.sny), torch.arange(O.snx), indexing="ij")
        sp = torch.stack([z, y, x], -1).view(-1, 3) * O.s_rev_vs + O.s_rev_origin
        cyl = occ_oracle.cartesian_cylinder_coords(occ_oracle.sphere_uvd2absxyz(sp[..., 2], sp[..., 1], sp[..., 0]))
    c, inds = O.point2coords_inrange(cyl, O.origin, O.pmax, O.max_grid, O.min_grid, O.vs)
    b = torch.full((cyl.shape[0],), -1, dtype=torch.int64)
    b[inds] = (c[..., 2] * O.ny + c[..., 1]) * O.nx + c[..., 0]
    n = int((a.view(-1).long() != b).sum())
    assert 1000 < n < 0.2 * a.numel(), n
