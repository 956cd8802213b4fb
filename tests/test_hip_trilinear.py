"""The ROI head's resident trilinear read-out (csrc/roi_pool.hip: btc_trilinear_corners / _gather / _scatter behind
conv_head.trilinear_splat_resident) against the densify-and-index formulation it replaces (conv_head.trilinear_readout, itself equal to
the reference's ConvHead on convhead.npz): kept points and features bit for bit, the gradient into the sparse features within fp32
summation-order noise and run-to-run identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(seed, n_vox=5000, q_scenes=600, P=96, C=128, B=2):
    from btcdet_amd import spconv
    g = torch.Generator().manual_seed(seed)
    shape = [5, 200, 176]
    cells = torch.stack([torch.randint(0, B, (n_vox,), generator=g), torch.randint(0, shape[0], (n_vox,), generator=g),
                         torch.randint(60, 140, (n_vox,), generator=g), torch.randint(40, 130, (n_vox,), generator=g)], dim=1)
    cells = torch.unique(cells, dim=0).int()
    feats = torch.relu(torch.randn(cells.shape[0], C, generator=g))
    feats[::37] = 0.0                                                   # rows that are entirely zero: present cells that read as empty
    x = spconv.SparseConvTensor(feats.to(DEV).requires_grad_(True), cells.to(DEV), shape, B)
    # lattice points around the occupied region, in world coordinates (stride 8 grid of 0.05 x 0.05 x 0.1 m voxels, KITTI range)
    pts = torch.stack([torch.rand(q_scenes * P, generator=g) * 40.0 + 14.0, torch.rand(q_scenes * P, generator=g) * 36.0 - 18.0,
                       torch.rand(q_scenes * P, generator=g) * 5.0 - 3.5], dim=1)
    return x, pts.to(DEV), P, q_scenes // B


def test_resident_read_out_equals_the_dense_formulation():
    from btcdet_amd import conv_head
    rng, vs, stride = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0], [0.05, 0.05, 0.1], [8, 8, 8]
    x, pts, P, per_scene = _case(3)
    B = 2
    # the formulation of rounds 1-3
    zyx = torch.stack([(pts[:, 2] - rng[2]) / vs[2] / stride[0] - 0.5, (pts[:, 1] - rng[1]) / vs[1] / stride[1] - 0.5,
                       (pts[:, 0] - rng[0]) / vs[0] / stride[2] - 0.5], dim=-1)
    scene = torch.arange(pts.shape[0] // P, device=DEV)
    b = (scene // per_scene)[:, None].expand(-1, P).reshape(-1)
    ref = conv_head.trilinear_readout(x, b, zyx)
    keep_ref = torch.nonzero((ref.abs() > 0).any(dim=-1))[:, 0]
    out_ref = ref[keep_ref]
    gsel = torch.randn(out_ref.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    (g_ref,) = torch.autograd.grad(out_ref, x.features, gsel, retain_graph=False)
    # resident
    grads = []
    for _ in range(2):
        keep, out = conv_head.trilinear_splat_resident(x, pts.contiguous(), P * per_scene, B, rng, vs, stride)
        assert torch.equal(keep, keep_ref) and 0.02 * pts.shape[0] < keep.numel() < 0.9 * pts.shape[0]
        assert torch.equal(out, out_ref)                                  # same products, same order of the sum: bit for bit
        (g,) = torch.autograd.grad(out, x.features, gsel)
        grads.append(g)
    assert torch.equal(grads[0], grads[1])                                # deterministic
    scale = float(g_ref.abs().max())
    assert float((grads[0] - g_ref).abs().max()) <= 2e-6 * scale          # the order of the per-row sum differs from index_put's
    assert float(grads[0][::37].abs().max()) > 0                          # all-zero rows are read, and receive their gradient, all the same


def test_all_zero_rows_and_empty_result():
    from btcdet_amd import conv_head, spconv
    shape = [5, 200, 176]
    x = spconv.SparseConvTensor(torch.zeros(4, 16, device=DEV), torch.tensor([[0, 1, 100, 80], [0, 2, 100, 80], [1, 1, 90, 70], [1, 4, 91, 70]],
                                                                             dtype=torch.int32, device=DEV), shape, 2)
    pts = torch.tensor([[32.0, 0.0, -2.0]] * 8, device=DEV)
    keep, out = conv_head.trilinear_splat_resident(x, pts, 4, 2, [0.0, -40.0, -3.0], [0.05, 0.05, 0.1], [8, 8, 8])
    assert keep.numel() == 0 and tuple(out.shape) == (0, 16)
