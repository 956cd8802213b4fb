"""Test-side restatement of the graph of VoxelBackBone8xOcc.forward for the configured options
(OCC_CONV_TYPE ['identity','maxpool'], OCC_CONV_EXECUTE [False, True], OUT_FEAT_TYPE [...,'big_bev_combine']), written
directly from /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:

  * forward                       :936-1019
  * sparse_cat (row alignment of the max-pool side branch with conv2's output rows)  :869-873
  * res_combine (down2 / down3 / squeezeBev / BEV gather / down_combine)             :905-918
  * compress_height               :920-933
  * layer tables (channels, kernels, strides, paddings, indice_keys)                 :656-700, :732-767, :831-847
  * HeightCompression.forward     backbones_2d/map_to_bev/height_compression.py:20-25

Nothing here imports btcdet_amd: rulebooks come from the C oracle (oracle.rulebook), features either from the oracle's
fp32 conv (oracle.conv_fwd / maxpool_fwd / dense: `forward_np`) or from a differentiable torch-CPU float64 gather-matmul over
the same oracle rulebooks (`forward_t64`, used for the gradients).  Test infrastructure only."""
import numpy as np
import torch

from oracle import oracle as orc

EPS = 1e-3
DET_SHAPE = [41, 1600, 1408]

# (state_dict prefix of the conv, rulebook name); BatchNorm of layer "a.b.0" is "a.b.1", of "conv1.0" is "conv1.1"
RULEBOOKS = [
    # name, input level, kernel, stride, padding, mode
    ("subm1", "l0", 3, 1, 0, orc.MODE_SUBM),
    ("spconv2", "l0", 3, 2, 1, orc.MODE_CONV),
    ("pool2", "l0", 3, 2, 1, orc.MODE_CONV),            # SparseMaxPool3d(3, stride 2, padding 1), :836 -- no indice_key
    ("subm2", "l1", 3, 1, 0, orc.MODE_SUBM),
    ("spconv3", "l1", 3, 2, 1, orc.MODE_CONV),
    ("subm3", "l2", 3, 1, 0, orc.MODE_SUBM),
    ("spconv4", "l2", 3, 2, (0, 1, 1), orc.MODE_CONV),
    ("subm4", "l3", 3, 1, 0, orc.MODE_SUBM),
    ("spconv_down2", "l3", (3, 1, 1), (2, 1, 1), 0, orc.MODE_CONV),
    ("subm_down2", "l4", (2, 1, 1), (2, 1, 1), 0, orc.MODE_CONV),
]
LEVEL_OF = {"spconv2": "l1", "spconv3": "l2", "spconv4": "l3", "spconv_down2": "l4", "subm_down2": "l5"}


def geometry(coords, shape=DET_SHAPE):
    """every rulebook the forward pass builds, from the oracle; -> dict name -> (out_idx, nbr_out, nbr_in), levels"""
    lv = {"l0": (np.ascontiguousarray(coords, np.int32), list(shape))}
    rb = {}
    for name, lin, k, s, p, mode in RULEBOOKS:
        idx, shp = lv[lin]
        o_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shp, k, s, p, 1, mode)
        rb[name] = (o_idx, nbr_out, nbr_in)
        if name in LEVEL_OF:
            lv[LEVEL_OF[name]] = (o_idx, [int(v) for v in osh])
    # the reference relies on this (sparse_cat, :869-873): the pool's output rows are the conv's output rows
    assert np.array_equal(rb["pool2"][0], rb["spconv2"][0])
    return rb, lv


class _Np(object):
    """fp32 evaluation with the oracle's conv"""

    def __init__(self, sd, train):
        self.sd, self.train = sd, train

    def conv(self, f, name, nbr_out):
        return orc.conv_fwd(f, self.sd[name + ".weight"], None, nbr_out)

    def bn_relu(self, f, name):
        w, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        if self.train:
            f64 = f.astype(np.float64)
            mean, var = f64.mean(0), f64.var(0)
        else:
            mean, var = self.sd[name + ".running_mean"].astype(np.float64), self.sd[name + ".running_var"].astype(np.float64)
        y = (f.astype(np.float64) - mean) / np.sqrt(var + EPS) * w + b
        return np.maximum(y, 0).astype(np.float32)

    def maxpool(self, f, nbr_out):
        return orc.maxpool_fwd(f, nbr_out)

    def cat(self, xs):
        return np.concatenate(xs, axis=1)

    def dense(self, f, idx, bs, shape):
        return orc.dense(f, idx, bs, shape)

    def bev_rows(self, bev2d, idx):
        return bev2d[idx[:, 0], :, idx[:, 2], idx[:, 3]]


class _T64(object):
    """differentiable float64 evaluation on the CPU over the oracle's rulebooks"""

    def __init__(self, params, train, sd, masks=None):
        self.p, self.train, self.sd, self.masks = params, train, sd, masks

    def conv(self, f, name, nbr_out):
        W = self.p[name + ".weight"]
        W = W.reshape(-1, W.shape[-2], W.shape[-1])
        out = f.new_zeros(nbr_out.shape[0], W.shape[-1])
        for k in range(nbr_out.shape[1]):
            o = np.nonzero(nbr_out[:, k] >= 0)[0]
            if o.size:
                out = out.index_add(0, torch.from_numpy(o), f[torch.from_numpy(nbr_out[o, k].astype(np.int64))] @ W[k])
        return out

    def bn_relu(self, f, name):
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        if self.train:
            y = torch.nn.functional.batch_norm(f, None, None, w, b, True, 0.0, EPS)
        else:
            rm = torch.from_numpy(self.sd[name + ".running_mean"]).double()
            rv = torch.from_numpy(self.sd[name + ".running_var"]).double()
            y = torch.nn.functional.batch_norm(f, rm, rv, w, b, False, 0.0, EPS)
        if self.masks is not None:      # the branch of the piecewise-linear ReLU the device took (see forward_t64)
            return y * torch.from_numpy(self.masks[name]).double()
        return torch.relu(y)

    def maxpool(self, f, nbr_out):
        pad = torch.cat([f, f.new_zeros(1, f.shape[1])])          # out initialised to 0, then max over the pairs (App. B.6)
        rows = torch.from_numpy(np.where(nbr_out >= 0, nbr_out, f.shape[0]).astype(np.int64))
        return pad[rows].max(1).values.clamp(min=0)

    def cat(self, xs):
        return torch.cat(xs, dim=1)

    def dense(self, f, idx, bs, shape):
        out = f.new_zeros(bs, shape[0], shape[1], shape[2], f.shape[1])
        i = torch.from_numpy(idx.astype(np.int64))
        out = out.index_put((i[:, 0], i[:, 1], i[:, 2], i[:, 3]), f)
        return out.permute(0, 4, 1, 2, 3).contiguous()

    def bev_rows(self, bev2d, idx):
        i = torch.from_numpy(idx.astype(np.int64))
        return bev2d[i[:, 0], :, i[:, 2], i[:, 3]]


def _graph(E, rb, lv, feats, occ_feats, bs):
    """the layer graph; E evaluates the primitives.  Line numbers: spconv_backbone.py"""
    def block(f, prefix, rbname):                  # post_act_block :7-43 = conv, BatchNorm1d, ReLU
        return E.bn_relu(E.conv(f, prefix + ".0", rb[rbname][1]), prefix + ".1")

    x1 = E.bn_relu(E.conv(feats, "conv1.0", rb["subm1"][1]), "conv1.1")                 # :957 (conv1 is a bare triple, :656-660)
    x1 = block(x1, "conv1_combine.0", "subm1")                                             # :969 (exec[0] False: no cat)
    x2 = block(x1, "conv2.0", "spconv2")                                                   # :970
    occ2 = E.maxpool(occ_feats, rb["pool2"][1])                                            # :972
    x2 = E.cat([x2, occ2])                                                                 # :974 sparse_cat
    x2 = block(block(x2, "conv2_combine.0", "subm2"), "conv2_combine.1", "subm2")          # :977
    x3 = block(x2, "conv3.0", "spconv3")                                                   # :978
    x3 = block(block(x3, "conv3_combine.0", "subm3"), "conv3_combine.1", "subm3")          # :985
    x4 = block(x3, "conv4.0", "spconv4")                                                   # :987
    x4 = block(block(x4, "conv4_combine.0", "subm4"), "conv4_combine.1", "subm4")          # :994
    out = E.bn_relu(E.conv(x4, "conv_out.0", rb["spconv_down2"][1]), "conv_out.1")        # :1003
    # res_combine :905-918
    d2 = block(block(x2, "down2.0", "spconv3"), "down2.1", "spconv4")                      # :907 reuses spconv3 / spconv4 rulebooks
    d3 = block(x3, "down3.0", "spconv4")                                                   # :908
    cat = E.cat([d2, d3, x4])                                                              # :909
    bev = block(out, "squeezeBev.0", "subm_down2")                                         # :911
    idx5, shp5 = lv["l5"]
    bev3d = E.dense(bev, idx5, bs, shp5)                                                   # :912 compress_height :930-932
    bev2d = bev3d.reshape(bev3d.shape[0], bev3d.shape[1] * bev3d.shape[2], bev3d.shape[3], bev3d.shape[4])
    cat = E.cat([cat, E.bev_rows(bev2d, lv["l3"][0])])                                     # :913-915
    xc = block(block(cat, "down_combine.0", "subm4"), "down_combine.1", "subm4")           # :916
    idx4, shp4 = lv["l4"]
    vol = E.dense(out, idx4, bs, shp4)                                                     # height_compression.py:21
    sf = vol.reshape(vol.shape[0], vol.shape[1] * vol.shape[2], vol.shape[3], vol.shape[4])  # :23
    return {"out": out, "out_indices": idx4, "out_shape": shp4, "x_combine": xc, "x_combine_indices": lv["l3"][0],
            "x_combine_shape": lv["l3"][1], "spatial_features": sf, "x_conv2": x2, "x_conv3": x3, "x_conv4": x4}


def forward_np(sd, rb, lv, feats, occ_feats, bs, train):
    return _graph(_Np(sd, train), rb, lv, np.ascontiguousarray(feats, np.float32), np.ascontiguousarray(occ_feats, np.float32), bs)


def forward_t64(sd, rb, lv, feats, occ_feats, bs, train, masks=None):
    """-> (outputs, params dict of float64 leaves, input leaf).  masks: {BatchNorm name: bool (N, C)} = which units were active
    in the run this one is compared with.  ReLU is not differentiable at 0: of the ~10^7 pre-activations of one pass a handful
    lie within float32 rounding of 0 and come out on the other side in float64, and one such flip moves a BatchNorm's dbeta
    (hence dx of every row of that channel, hence every gradient upstream) by far more than rounding.  Imposing the masks
    makes both runs differentiate the same linear branch, so the comparison measures arithmetic, not branch choice."""
    params = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in sd.items()
              if k.endswith(".weight") or k.endswith(".bias")}
    x = torch.from_numpy(np.ascontiguousarray(feats)).double().requires_grad_(True)
    occ = torch.from_numpy(np.ascontiguousarray(occ_feats)).double()
    return _graph(_T64(params, train, sd, masks), rb, lv, x, occ, bs), params, x
