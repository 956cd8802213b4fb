"""Sparse 3-D backbones of BtcDet's two branches, built on btcdet_amd.spconv.

Mirrors (same constructor protocol, attribute / parameter names, layer graph, channel / stride /
padding / indice_key tables -- so reference checkpoints load key-for-key):
  * ``post_act_block``      /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:7-43
  * ``VoxelBackBoneDeconv`` spconv_backbone.py:91-203   (occupancy branch)
  * ``VoxelBackBone8xOcc``  spconv_backbone.py:630-1019 (detection branch)
The graphs are written as layer tables; every sparse layer is one fused HIP launch (sparse_conv.hip).
"""
import os
from functools import partial

import torch
import torch.nn as nn

from . import spconv


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None, defaultvalue=1.0, activation=nn.ReLU):
    """conv (+ norm + activation) as one SparseSequential; conv_type selects the sparse layer."""
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'subm2d':
        conv = spconv.SubMConv2d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == 'fixspconv':
        conv = fixSparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                               indice_key=indice_key, defaultvalue=defaultvalue)
        conv.requires_grad_(False)
    elif conv_type == 'spdeconv':
        conv = spconv.SparseConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                            bias=False, indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    elif conv_type == 'submbias':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=True, indice_key=indice_key)
    elif conv_type == 'maxpool':
        conv = spconv.SparseMaxPool3d(kernel_size, stride=stride, padding=padding)
    else:
        raise NotImplementedError(conv_type)
    if norm_fn is not None:
        return spconv.SparseSequential(conv, norm_fn(out_channels), activation())
    return spconv.SparseSequential(conv)


def _feature_dtype(model_cfg):
    """FEATURE_DTYPE: bf16 -> the sparse tensors between layers are bfloat16 (BASELINE.json configs[2] "bf16 features");
    weights, accumulation, BatchNorm statistics and every dense map stay fp32.  Not a key of the reference's yaml."""
    v = str(model_cfg.get("FEATURE_DTYPE", "fp32")).lower()
    if v in ("bf16", "bfloat16"):
        return torch.bfloat16
    if v in ("fp32", "float32"):
        return None
    raise ValueError("FEATURE_DTYPE must be fp32 or bf16, got %r" % v)


class fixSparseConv3d(spconv.SparseConv3d):
    """constant-weight sparse conv (spconv_backbone.py:45-48)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None,
                 defaultvalue=1.0):
        super(fixSparseConv3d, self).__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                              bias=bias, indice_key=indice_key)
        self.weight.data.fill_(defaultvalue)


def _to_feature_dtype(x, dtype):
    """the sparse tensor with its features in the backbone's FEATURE_DTYPE (a no-op for fp32 models and once the features are there)"""
    if dtype is not None and x.features is not None and x.features.dtype != dtype and x.features.shape[1] % 16 == 0:
        x.features = x.features.to(dtype)
    return x


def _strided(seq):
    """first sparse layer of a SparseSequential stage if it builds its own (strided / transposed / dilating) rulebook"""
    m = seq[0]
    while isinstance(m, spconv.SparseSequential):
        m = m[0]
    return m if isinstance(m, spconv.SparseConvolution) and not m.subm and not m.inverse else None


def _chain_lookahead(stages):
    """stage i's rulebook-building layer announces stage i+1's: its row count starts on the side stream as soon as the level
    it consumes exists (spconv/ops.py LOOKAHEAD).  Plain attribute, not a submodule registration."""
    layers = [l for l in (_strided(s) for s in stages) if l is not None]
    for a, b in zip(layers[:-1], layers[1:]):
        a.__dict__['lookahead'] = (b,)
    return layers[0] if layers else None


def _seq(specs, norm_fn):
    """specs: list of (cin, cout, k, dict(kwargs)) -> SparseSequential of post_act_blocks"""
    return spconv.SparseSequential(*[post_act_block(ci, co, k, norm_fn=norm_fn, **kw) for ci, co, k, kw in specs])


class VoxelBackBoneDeconv(nn.Module):
    """Occupancy-branch encoder/decoder: [9,157,209] -> /2 -> /4 -> x2 -> x2, 32 output channels."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.y_shift = model_cfg.get("SHIFT", 0)
        self.feature_dtype = _feature_dtype(model_cfg)
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = grid_size[::-1]  # numpy view of the dataset's grid, as in the reference (App. D.10)
        self.sparse_shape[1] += self.y_shift * 2
        c = [16, 32, 64]
        sp = dict(padding=1, conv_type='spconv')
        self.conv1 = spconv.SparseSequential(post_act_block(input_channels, c[0], 3, norm_fn=norm_fn, indice_key='spconv1', **sp))
        self.conv2 = _seq([(c[0], c[1], 3, dict(stride=2, indice_key='spconv2', **sp)),
                           (c[1], c[1], 3, dict(padding=1, indice_key='subm2'))], norm_fn)
        self.conv3 = _seq([(c[1], c[2], 3, dict(stride=2, indice_key='spconv3', **sp)),
                           (c[2], c[2], 3, dict(padding=1, indice_key='subm3'))], norm_fn)
        self.deconv4 = _seq([(c[2], c[1], 3, dict(stride=2, padding=1, indice_key='spconv4', conv_type='spdeconv')),
                             (c[1], c[1], 3, dict(padding=1, indice_key='subm4'))], norm_fn)
        self.deconv5 = _seq([(c[1], c[1], 3, dict(stride=2, padding=1, indice_key='spconv5', conv_type='spdeconv')),
                             (c[1], c[1], 3, dict(padding=1, indice_key='subm5'))], norm_fn)
        self.num_point_features = c[1]
        _chain_lookahead([self.conv1, self.conv2, self.conv3, self.deconv4, self.deconv5])

    def forward(self, batch_dict):
        voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords'].int()
        if self.feature_dtype is not None and voxel_features.shape[1] % 16 == 0:
            voxel_features = voxel_features.to(self.feature_dtype)
        # (a 4-channel input stays fp32: the first layer computes in fp32 either way -- rounding its input to bf16, widening it again
        # and rounding the result were three launches for nothing; the cast follows the first stage, _forward_stages)
        if self.y_shift > 0:
            voxel_features, voxel_coords = self.add_shift(voxel_features, voxel_coords)
        x = spconv.SparseConvTensor(features=voxel_features, indices=voxel_coords, spatial_shape=self.sparse_shape,
                                    batch_size=batch_dict['batch_size'])
        ready = batch_dict.pop('occ_geometry', None)
        if ready is not None and ready[0] is voxel_coords:  # rulebooks of this very coordinate tensor (prefetch_geometry)
            x.indice_dict = ready[1]
        x = self._forward_stages(x)
        if self.y_shift > 0:
            x = self.remove_shift(x)
        batch_dict.update({'encoded_spconv_tensor': x, 'encoded_spconv_tensor_stride': 1})
        return batch_dict

    def _forward_stages(self, x):
        """conv1 .. deconv5, one compiled chain call per stage when its rulebooks are at hand (prefetch_geometry).  (All five stages as
        ONE call measured no better -- 430 / 453 / 440 scenes/s against 438 / 458 / 455 per stage, round 4 -- and is not kept.)"""
        for stage in (self.conv1, self.conv2, self.conv3, self.deconv4, self.deconv5):
            x = stage(x)
            x = _to_feature_dtype(x, self.feature_dtype)
        return x

    def prefetch_geometry(self, batch_dict, head=None):
        """every rulebook forward (and the occupancy head) will need, built from the voxel coordinates alone -- they do not
        depend on the weights, so a training loop can build them for the NEXT batch while this batch's backward runs
        (BtcHotPath.prepare).  Stored as batch_dict['occ_geometry'] = (coords, indice_dict); forward picks it up when it is
        handed the same coordinate tensor."""
        if self.y_shift > 0:
            return batch_dict
        voxel_coords = batch_dict['voxel_coords'].int()
        if not voxel_coords.is_cuda:
            return batch_dict
        from .spconv import ops as sp_ops
        stages = (self.conv1, self.conv2, self.conv3, self.deconv4, self.deconv5)
        bs = batch_dict['batch_size']
        merged_head = head is not None and hasattr(head, "_merge_ok") and head._merge_ok()
        if sp_ops.fast() is not None and (head is None or merged_head or hasattr(head, "conv_cls")):
            # one call of the compiled binding for all rulebooks (spconv/geometry.py); the plan is fixed per (model, batch size)
            from .spconv.geometry import GeometryPlan, flatten_convs
            plans = self.__dict__.setdefault("_geometry_plans", {})
            key = (int(bs), id(head))
            plan = plans.get(key)
            if plan is None:
                convs = flatten_convs(*stages)
                if head is not None:
                    convs += flatten_convs(head.conv_cls) + (flatten_convs(head.conv_res) if getattr(head, "reg", False) else [])
                plan = plans[key] = GeometryPlan(convs, self.sparse_shape, bs)
            indice_dict = {}
            plan.run(voxel_coords, indice_dict)
            batch_dict['occ_geometry'] = (voxel_coords, indice_dict)
            return batch_dict
        x = spconv.SparseConvTensor(features=None, indices=voxel_coords, spatial_shape=self.sparse_shape, batch_size=bs)
        for stage in stages:
            x = stage.forward_geometry(x)
        if head is not None and hasattr(head, "forward_geometry"):
            head.forward_geometry(x)
        batch_dict['occ_geometry'] = (voxel_coords, x.indice_dict)
        return batch_dict

    # azimuth wrap-around padding (SHIFT, off in the configured model)
    def add_shift(self, voxel_features, voxel_coords):
        y_max = self.sparse_shape[1] - 2 * self.y_shift
        left = voxel_coords[..., 2] < self.y_shift
        right = voxel_coords[..., 2] >= (y_max - self.y_shift)
        lc, rc = voxel_coords[left, :].clone(), voxel_coords[right, :].clone()
        rc[..., 2] -= y_max
        lc[..., 2] += y_max
        feats = torch.cat([voxel_features[right, :], voxel_features, voxel_features[left, :]], dim=0)
        coords = torch.cat([rc, voxel_coords, lc], dim=0)
        coords[..., 2] += self.y_shift
        return feats, coords

    def remove_shift(self, x):
        y_max = self.sparse_shape[1] - 2 * self.y_shift
        x.indices[..., 2] -= self.y_shift
        keep = (x.indices[..., 2] >= 0) & (x.indices[..., 2] < y_max)
        x.features, x.indices = x.features[keep, :], x.indices[keep, :]
        x.spatial_shape[1] -= self.y_shift * 2
        return x


DET_GEOMETRY_WALK = True  # VoxelBackBone8xOcc._walk_geometry (False: the layers build their rulebooks one by one; tests flip it)
FAST_STAGES = True        # VoxelBackBone8xOcc._stage: stages straight into the compiled chain call


class VoxelBackBone8xOcc(nn.Module):
    """Detection-branch 8x backbone that also consumes the occupancy code channels."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = grid_size[::-1] + [1, 0, 0]
        self.occ_conv_type = self.model_cfg.OCC_CONV_TYPE
        self.occ_conv_exec = self.model_cfg.OCC_CONV_EXECUTE
        self.out_feat_type = getattr(self.model_cfg, "OUT_FEAT_TYPE", ["None", "None", "None", "None", "combine"])
        self.out_att = getattr(self.model_cfg, "OCC_ATT", [False, False, False, False])
        c = [16, 32, 64, 64, 128]
        self.occ_code_num = input_channels - kwargs["original_num_rawpoint_features"]
        add = [self.occ_code_num if t else 0 for t in self.occ_conv_exec] + [0] * (4 - len(self.occ_conv_exec))

        for i in range(1, len(self.occ_conv_exec)):
            self._build_occ_net(self.occ_conv_type[i], i)
        for i in range(len(self.occ_conv_exec)):
            if self.out_att[i]:
                ch = c[i] + add[i]
                setattr(self, 'att_conv%d' % (i + 1), spconv.SparseSequential(post_act_block(
                    ch, ch, 3, norm_fn=norm_fn, stride=1, padding=1, indice_key='subm%d' % (i + 1), activation=nn.LeakyReLU)))

        self.conv1 = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, c[0], 3, padding=1, bias=False, indice_key='subm1'), norm_fn(c[0]), nn.ReLU())
        self.conv1_combine = _seq([(c[0] + add[0], c[0], 3, dict(padding=1, indice_key='subm1'))], norm_fn)
        down = [None, dict(stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
                dict(stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
                dict(stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv')]
        for lvl in (1, 2, 3):
            setattr(self, 'conv%d' % (lvl + 1), _seq([(c[lvl - 1], c[lvl], 3, down[lvl])], norm_fn))
            key = 'subm%d' % (lvl + 1)
            setattr(self, 'conv%d_combine' % (lvl + 1),
                    _seq([(c[lvl] + add[lvl], c[lvl], 3, dict(padding=1, indice_key=key)),
                          (c[lvl], c[lvl], 3, dict(padding=1, indice_key=key))], norm_fn))
        last_pad = self.model_cfg.get('last_pad', 0)
        self.feature_dtype = _feature_dtype(model_cfg)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(c[3], c[4], (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(c[4]), nn.ReLU())
        self.num_point_features = 128

        for i in range(4):
            if self.out_feat_type[i] == "2D":
                ch = c[i] * [41, 21, 11, 5][i]
                setattr(self, 'squeeze_z_conv%d' % (i + 1), spconv.SparseSequential(post_act_block(
                    ch, ch // 2, 3, norm_fn=norm_fn, padding=1, indice_key='submsqueez%d' % (i + 1), conv_type='subm2d')))
        self._build_combine_net(norm_fn, c, self.out_feat_type[4])
        stages = [self.conv2, self.conv3, self.conv4, self.conv_out]
        if getattr(self, "squeezeBev", None) is not None:
            stages.append(self.squeezeBev)
        self.__dict__['_first_strided'] = _chain_lookahead(stages)
        # the strided levels of the rulebook walk built beside the first stage on a side stream (True), or the whole walk first with one
        # blocking read-back (False).  Per instance: a schedule that already runs the branch on a stream of its own switches it off
        # (HotPathTrainer, pipelined: a fifth active stream costs 1.8 ms per step there).
        self.walk_async = True

    def _walk_geometry(self, coords, bs, indice_dict, allow_async=True):
        """all rulebooks of the main chain (subm1, spconv2, subm2, ... spconv_down2, subm_down2) in one call of the compiled
        binding before the first layer runs (spconv/geometry.py): every stage then finds its rulebooks ready and runs as one
        compiled call (SparseSequential._chain_plan) instead of ~100 us of Python per layer; the side-branch pools and the
        down2 / down3 / down_combine layers reuse these rulebooks through the geometry cache / their indice_keys as before.
        False when the compiled binding is not in use (the layers then build their rulebooks one by one, with lookahead)."""
        from .spconv import ops as sp_ops
        if not (DET_GEOMETRY_WALK and coords.is_cuda and sp_ops.fast() is not None and sp_ops.CAPTURE is None):
            return False
        from .spconv.geometry import GeometryPlan, flatten_convs
        plans = self.__dict__.setdefault("_geometry_plans", {})
        plan = plans.get(int(bs))
        if plan is None:
            # every sparse conv of forward() in execution order: the main chain builds, the side layers (conv1_combine, down2, down3,
            # down_combine) reuse by indice_key -- the plan then hands forward() one rulebook per layer (stage_rulebooks)
            stages = [self.conv1, self.conv1_combine, self.conv2, self.conv2_combine, self.conv3, self.conv3_combine, self.conv4, self.conv4_combine,
                      self.conv_out]
            if getattr(self, "squeezeBev", None) is not None:
                stages.append(self.squeezeBev)
            if getattr(self, "down3", None) is not None:
                stages += [self.down2, self.down3, self.down_combine]
            plan = plans[int(bs)] = GeometryPlan(flatten_convs(*stages), self.sparse_shape, bs)
            offs, pos = {}, 0
            for st in stages:
                n = len(flatten_convs(st))
                offs[id(st)] = (pos, pos + n)
                pos += n
            plan.stage_slices = offs
        if allow_async and self.walk_async and sp_ops.PROFILE is None and plan.entries[0][0] == 0:
            # The first stage (conv1, conv1_combine) only needs the level-0 submanifold rulebook, which needs no read-back: build it
            # alone, fork the rest of the walk (the strided levels and the read-back of their row counts) onto a side stream, and
            # let the caller run the first stage before it joins (forward -> _finish_walk).  The walk's ~0.3 ms of kernels and its
            # read-back bubble then sit beside the first stage's launches instead of in front of them.
            conv0 = plan.convs[0]
            rb0 = sp_ops.build_rulebook_g(coords, bs, conv0._geometry(self.sparse_shape))
            if conv0.indice_key is not None:
                indice_dict[conv0.indice_key] = rb0
            indice_dict.setdefault("__geometry_cache__", {})[conv0._gkey(coords, self.sparse_shape)] = (rb0, coords)
            return (plan, plan.start(coords, side_stream=True), {0: rb0}, coords)
        return ("done", plan, plan.run(coords, indice_dict))

    @staticmethod
    def _finish_walk(walk, indice_dict):
        """-> (plan, one rulebook per sparse conv of the plan) once the walk is complete, else None"""
        if isinstance(walk, tuple) and walk[0] == "done":
            return walk[1], walk[2]
        if isinstance(walk, tuple):
            plan, handle, have, coords = walk
            return plan, plan.finish(handle, coords, indice_dict, have)
        return None

    def _stage(self, stage, x, ready):
        """run a SparseSequential stage.  With the walk's rulebooks at hand (ready = (plan, rulebooks)) a pure conv -> BatchNorm -> ReLU
        stage goes straight into ONE call of the compiled binding (SparseSequential._run_chain) -- no module call, no rulebook
        look-ups, no per-layer Python: the training thread's forward pass is bound by exactly that"""
        from .spconv import fused_bn as sp_fused_bn, modules as sp_modules, ops as sp_ops
        if ready is not None and sp_modules.CHAIN_LAYERS and sp_modules.FUSE_CONV_BN and sp_modules.FUSE_BN_RELU and sp_ops.PROFILE is None \
                and sp_ops.CAPTURE is None and sp_ops.NATIVE_AUTOGRAD and sp_ops.fast() is not None:   # (the conditions of SparseSequential's own chain path)
            plan, rbs = ready
            sl = plan.stage_slices.get(id(stage))
            triples = stage.__dict__.get("_chain_triples", False)
            if triples is False:
                stage._chain_plan(x)                      # derives and caches the stage's (conv, bn, relu) triples
                triples = stage.__dict__.get("_chain_triples", None)
            f = x.features
            if sl is not None and triples and len(triples) == sl[1] - sl[0] and f.is_cuda and f.shape[0] > 0 and \
                    all(sp_fused_bn.fusable(bn) for _, bn, _ in triples) and \
                    (f.dtype == torch.float32 or all(c.in_channels % 16 == 0 and c.out_channels % 16 == 0 for c, _, _ in triples)):
                mine = rbs[sl[0]:sl[1]]
                if all(rb is not None and rb.n_out > 0 for rb in mine):
                    steps = [(c, bn, relu, rb, False) for (c, bn, relu), rb in zip(triples, mine)]
                    return stage._run_chain(x, steps, mine[-1].out_indices, plan.entries[sl[1] - 1][4])
        return stage(x)

    def _stages(self, stages, x, ready):
        """consecutive stages with nothing between them, one compiled chain call each"""
        for stage in stages:
            x = self._stage(stage, x, ready)
        return x

    def _build_occ_net(self, kind, i):
        """occupancy-code side branch at level i (1..3): maxpool / learned / fixed-mean / avg (spconv_backbone.py:793-866)"""
        n = self.occ_code_num
        key = 'spconv%d' % (i + 1)
        pad = 1 if i < 3 else (1, 1, 1)
        kw = {'maxpool': dict(k=3, conv_type='maxpool'), 'weight': dict(k=3, conv_type='spconv'),
              'fix': dict(k=3, conv_type='fixspconv', defaultvalue=1.0 / 27),
              'avgpool': dict(k=2, conv_type='fixspconv', defaultvalue=1)}[kind]
        k = kw.pop('k')
        setattr(self, 'occ_conv%d' % (i + 1), spconv.SparseSequential(
            post_act_block(n, n, k, norm_fn=None, stride=2, padding=pad, indice_key=key, **kw)))

    def _build_combine_net(self, norm_fn, c, comb_type):
        sp = dict(stride=2, conv_type='spconv')
        self.down2 = _seq([(c[1], c[1], 3, dict(padding=1, indice_key='spconv3', **sp)),
                           (c[1], c[2], 3, dict(padding=(0, 1, 1), indice_key='spconv4', **sp))], norm_fn)
        self.down3 = _seq([(c[2], c[2], 3, dict(padding=(0, 1, 1), indice_key='spconv4', **sp))], norm_fn)
        cat = c[2] * 2 + c[3]
        if comb_type == "big_combine":
            self.down_combine = _seq([(cat, c[3] * 2, 3, dict(padding=1, indice_key='subm4')),
                                      (c[3] * 2, c[3] * 2, 3, dict(padding=1, indice_key='subm4'))], norm_fn)
        elif comb_type == "combine":
            self.down_combine = _seq([(cat, c[3] * 2, 3, dict(padding=1, indice_key='subm4')),
                                      (c[3] * 2, c[3] * 2, 3, dict(stride=[1, 2, 2], padding=(1, 1, 1), indice_key='spconv5',
                                                                   conv_type='spconv')),
                                      (c[3] * 2, c[3] * 2, 3, dict(padding=1, indice_key='subm5'))], norm_fn)
        elif comb_type == "big_bev_combine":
            self.squeezeBev = _seq([(c[4], c[3], (2, 1, 1), dict(stride=(2, 1, 1), padding=0, indice_key='subm_down2',
                                                                  conv_type='spconv'))], norm_fn)
            self.down_combine = _seq([(cat + c[3], c[3] * 2, 3, dict(padding=1, indice_key='subm4')),
                                      (c[3] * 2, c[3] * 2, 3, dict(padding=1, indice_key='subm4'))], norm_fn)

    @staticmethod
    def sparse_cat(input_lst, pad=False):
        # rows of both tensors are aligned because both rulebooks emit outputs in (b,z,y,x) order
        xrep, xocc = input_lst
        if pad:   # the consumer is a sparse conv: the zero channels it pads its input to (34 -> 64) come out of the same launch
            from .spconv import ops as sp_ops
            xrep.features = sp_ops.cat_features(xrep.features, xocc.features)
        else:
            xrep.features = torch.cat((xrep.features, xocc.features.to(xrep.features.dtype)), dim=1)
        return xrep

    @staticmethod
    def apply_att(x, att_conv):
        a = att_conv(x)
        x.features = x.features * a.features + x.features
        return x

    def suqeeze(self, feat, i, kind):
        if kind == "None":
            return None
        if kind == "3D":
            return feat
        conv = getattr(self, "squeeze_z_conv{}".format(i), None)
        dense = feat.dense()
        B, C, Z, Y, X = list(dense.shape)
        inds = torch.unique(torch.cat([feat.indices[..., 0:1], feat.indices[..., 2:]], dim=-1), dim=0).long()
        pix = dense.reshape(B, C * Z, Y, X)[inds[..., 0], :, inds[..., 1], inds[..., 2]]
        return conv(spconv.SparseConvTensor(features=pix, indices=inds.int(), spatial_shape=[Y, X], batch_size=B)).dense()

    @staticmethod
    def compress_height(t):
        d = t.dense()
        N, C, D, H, W = d.shape
        return d.view(N, C * D, H, W)

    def res_combine(self, x2, x3, x4, bev, out_feat_type="combine", ready=None):
        if getattr(self, "down3", None) is None:
            return None
        x2, x3 = self._stage(self.down2, x2, ready), self._stage(self.down3, x3, ready)
        if out_feat_type == "big_bev_combine":
            bev2d = self.compress_height(self._stage(self.squeezeBev, bev, ready))
            x4.features = torch.cat((x2.features, x3.features, x4.features,
                                     _BevGather.apply(bev2d, x4.indices, tuple(x4.spatial_shape), x4.batch_size).to(x4.features.dtype)), dim=1)
        else:
            x4.features = torch.cat((x2.features, x3.features, x4.features), dim=1)
        return self._stage(self.down_combine, x4, ready)

    @staticmethod
    def _coords_i32(batch_dict):
        from .vfe import i32_twin
        coords = batch_dict['voxel_coords']
        tw = i32_twin(batch_dict, coords)      # (PassOccVox's int64 coordinates come with an int32 twin: no conversion launch)
        return tw if tw is not None else coords.int()

    def prefetch_geometry(self, batch_dict):
        """every rulebook of this backbone for batch_dict['voxel_coords'] NOW -- by whoever has just produced the coordinates
        (BtcHotPath.forward_occ, right behind PassOccVox): the walk is a function of the coordinates alone, its ~0.3 ms of host time
        (one compiled call with a blocking read-back of the level sizes) is then off the thread that runs this branch's forward --
        under the pipelined schedule the training thread, whose host time IS the step (BTC_TRAINER_TIMING=1: det_forward 2.6 of
        4.6 ms, never waiting for anybody) -- and on the occupancy branch's worker, which has a millisecond to spare.
        forward() picks the result up if it is given the very same coordinate tensor."""
        coords = self._coords_i32(batch_dict)
        if not coords.is_cuda or coords.shape[0] == 0:
            return batch_dict
        indice_dict = {}
        walk = self._walk_geometry(coords, batch_dict['batch_size'], indice_dict, allow_async=False)
        if isinstance(walk, tuple) and walk[0] == "done":
            batch_dict['det_geometry'] = (coords, indice_dict, walk)
        return batch_dict

    def forward(self, batch_dict):
        coords = self._coords_i32(batch_dict)
        feats = batch_dict['voxel_features']
        if self.feature_dtype is not None and feats.shape[1] % 16 == 0:
            feats = feats.to(self.feature_dtype)     # (a 6-channel input stays fp32 through conv1, see VoxelBackBoneDeconv.forward)
        bs = batch_dict['batch_size']
        x = spconv.SparseConvTensor(features=feats, indices=coords, spatial_shape=self.sparse_shape, batch_size=bs)
        pre = batch_dict.pop('det_geometry', None)
        if pre is not None and pre[0] is coords:     # rulebooks of this very coordinate tensor (prefetch_geometry)
            x.indice_dict = pre[1]
            walk = pre[2]
        else:
            walk = self._walk_geometry(coords, bs, x.indice_dict)
        if not walk and self._first_strided is not None:
            # conv2's row count runs beside conv1 (rulebook lookahead, spconv/ops.py)
            self._first_strided.prefetch(coords, self.sparse_shape, bs, x.indice_dict)
        n_occ = len(self.occ_conv_exec)
        ready = None
        if FAST_STAGES and isinstance(walk, tuple) and walk[0] == "done":   # the blocking walk: every rulebook is there already
            ready = self._finish_walk(walk, x.indice_dict)
        merge_first = ready is not None and not (n_occ > 0 and self.occ_conv_exec[0])      # nothing between conv1 and conv1_combine
        x1 = _to_feature_dtype(self._stage(self.conv1, x, ready), self.feature_dtype)
        if merge_first:
            x1 = self._stage(self.conv1_combine, x1, ready)
        occ = None
        if n_occ > 0:
            occ = spconv.SparseConvTensor(features=batch_dict["occ_voxel_features"], indices=coords,
                                          spatial_shape=self.sparse_shape, batch_size=bs)
            # rulebooks are pure functions of (indices, geometry): the side branch's pools share the main branch's builds
            occ.indice_dict["__geometry_cache__"] = x.indice_dict.setdefault("__geometry_cache__", {})
            if self.occ_conv_exec[0]:
                x1 = self.sparse_cat([x1, occ], pad=not self.out_att[0])
                if self.out_att[0]:
                    x1 = self.apply_att(x1, self.att_conv1)
        if not merge_first:
            x1 = self._stage(self.conv1_combine, x1, ready)
        if ready is None:   # the strided levels' rulebooks, built beside the first stage
            ready = self._finish_walk(walk, x.indice_dict)
            if not FAST_STAGES:
                ready = None
        levels = [x1]
        cur = x1
        for lvl in (1, 2, 3):
            if not (n_occ > lvl):   # no occupancy side branch at this level: conv{l} and conv{l}_combine back to back
                cur = self._stages([getattr(self, 'conv%d' % (lvl + 1)), getattr(self, 'conv%d_combine' % (lvl + 1))], cur, ready)
                levels.append(cur)
                continue
            cur = self._stage(getattr(self, 'conv%d' % (lvl + 1)), cur, ready)
            if n_occ > lvl:
                occ = getattr(self, 'occ_conv%d' % (lvl + 1))(occ)
                if self.occ_conv_exec[lvl]:
                    cur = self.sparse_cat([cur, occ], pad=not self.out_att[lvl])
                    if self.out_att[lvl]:
                        cur = self.apply_att(cur, getattr(self, 'att_conv%d' % (lvl + 1)))
            cur = self._stage(getattr(self, 'conv%d_combine' % (lvl + 1)), cur, ready)
            levels.append(cur)
        x1, x2, x3, x4 = levels
        out = self._stage(self.conv_out, x4, ready)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        # active rows per level of this batch (host integers the rulebook walk has read back already; bench.py reports them)
        batch_dict['__level_rows__'] = {'det_L0': int(x1.features.shape[0]), 'det_L1': int(x2.features.shape[0]),
                                        'det_L2': int(x3.features.shape[0]), 'det_L3': int(x4.features.shape[0]),
                                        'det_out': int(out.features.shape[0])}
        batch_dict.update({'multi_scale_3d_features': {
            'x_conv1': self.suqeeze(x1, 1, self.out_feat_type[0]),
            'x_conv2': self.suqeeze(x2, 2, self.out_feat_type[1]),
            'x_conv3': self.suqeeze(x3, 3, self.out_feat_type[2]),
            'x_conv4': self.suqeeze(x4, 4, self.out_feat_type[3]),
            'x_combine': self.res_combine(x2, x3, x4, out, out_feat_type=self.out_feat_type[4], ready=ready),
        }})
        return batch_dict


# ----------------------------------------------------------------------------------------------------------------------
# The backbone variants the reference can reach by changing BACKBONE_3D.NAME: residual blocks,
# lateral "combine" decoders and the inverse-convolution decoder.  Same constructor arguments, parameter names and
# batch_dict keys as /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:50-88,226-627.
# ----------------------------------------------------------------------------------------------------------------------
class _BevGather(torch.autograd.Function):
    """rows[r] = bev2d[b_r, :, y_r, x_r] for the active cells [b,z,y,x] of a sparse tensor (spconv_backbone.py:1045-1048 writes
    this as advanced indexing).  Same forward; the backward of advanced indexing is index_put_(accumulate=True) -- ~20 launches
    on ROCm (index linearisation, a radix sort, the segmented accumulation) -- and here the gradient rows are scattered into
    the dense (B, C, D, H, W) volume (one launch, the .dense() kernel) and summed over D."""

    @staticmethod
    def forward(ctx, bev2d, indices, spatial_shape, batch_size):
        inds = indices.long()
        ctx.save_for_backward(indices)
        ctx.meta = (spatial_shape, batch_size, bev2d.dtype)
        return bev2d[inds[:, 0], :, inds[:, 2], inds[:, 3]]

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        spatial_shape, batch_size, dtype = ctx.meta
        vol = spconv.SparseConvTensor(grad.contiguous(), indices, list(spatial_shape), batch_size).dense()
        return vol.sum(2).to(dtype), None, None, None


class SparseBasicBlock(spconv.SparseModule):
    """two SubM 3x3x3 convs (with bias) + BatchNorm, identity shortcut, ReLU (spconv_backbone.py:50-88)"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x.features if self.downsample is None else self.downsample(x).features
        out = self.conv1(x)
        out.features = _bn_act(self.bn1, out.features, True)
        out = self.conv2(out)
        out.features = torch.relu(_bn_act(self.bn2, out.features, False) + identity)
        return out


def _bn_act(bn, feats, relu):
    """BatchNorm1d (+ ReLU) over sparse features through the fused HIP kernels when they apply"""
    from .spconv import fused_bn
    if feats.is_cuda and feats.dim() == 2 and fused_bn.is_sync(bn):     # --sync_bn: statistics over all ranks' rows
        return fused_bn.sync_batch_norm_relu(bn, feats, relu)
    if feats.is_cuda and feats.dim() == 2 and feats.shape[0] > 0 and fused_bn.fusable(bn):
        return fused_bn.batch_norm_relu(bn, feats, relu)
    y = bn(feats)
    return torch.relu(y) if relu else y


def _lateral_combine(x_lateral, x_bottom, conv_expnd, comb_conv):
    """decoder merge of the Res backbones (spconv_backbone.py:306-319): the lateral tensor goes through conv_expnd, its rows
    are placed at the rows of x_bottom that hold the same cell (rank of the cell among the sorted unique cells of both --
    which is the row of x_bottom because strided / transposed rulebooks emit rows in sorted cell order), absent cells get
    zeros, and the concatenation goes through comb_conv"""
    x_expnd = conv_expnd(x_lateral)
    M = x_bottom.features.shape[0]
    ind_all = torch.cat([x_bottom.indices, x_expnd.indices], dim=0)
    _, rinds = torch.unique(ind_all, dim=0, return_inverse=True)
    pad = torch.zeros((M, x_expnd.features.shape[1]), dtype=x_bottom.features.dtype, device=x_bottom.features.device)
    pad[rinds[M:], :] = x_expnd.features.to(pad.dtype)
    x_bottom.features = torch.cat([x_bottom.features, pad], dim=-1)
    return comb_conv(x_bottom)


class _ResDecoderBase(nn.Module):
    """shared forward of VoxelBackBoneDeconvRes / VoxelBackBoneInverseRes (spconv_backbone.py:322-381,478-527)"""

    def forward(self, batch_dict):
        feats, coords = batch_dict['voxel_features'], batch_dict['voxel_coords'].int()
        x = spconv.SparseConvTensor(features=feats, indices=coords, spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x2d = _lateral_combine(x2, self.deconv2(x3), self.conv22, self.comb_conv2)
        x1d = _lateral_combine(x1, self.deconv1(x2d), self.conv11, self.comb_conv1)
        batch_dict.update({'encoded_spconv_tensor': x1d, 'encoded_spconv_tensor_stride': 1})
        return batch_dict


class VoxelBackBoneDeconvRes(_ResDecoderBase):
    """encoder [/1, /2, /4] + transposed-conv decoder with lateral merges (spconv_backbone.py:226-381)"""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = grid_size[::-1]
        c = [16, 32, 64]
        sp, dc = dict(conv_type='spconv'), dict(conv_type='spdeconv')
        self.conv1 = _seq([(input_channels, c[0], 3, dict(padding=1, indice_key='spconv1', **sp))], norm_fn)
        self.conv2 = _seq([(c[0], c[1], 3, dict(stride=2, padding=1, indice_key='spconv2', **sp)),
                           (c[1], c[1], 3, dict(padding=1, indice_key='subm2'))], norm_fn)
        self.conv3 = _seq([(c[1], c[2], 3, dict(stride=2, padding=1, indice_key='spconv3', **sp)),
                           (c[2], c[2], 3, dict(padding=1, indice_key='subm3'))], norm_fn)
        self.conv22 = _seq([(c[1], c[1], 3, dict(stride=1, padding=1, indice_key='spconv22', **sp))], norm_fn)
        self.deconv2 = _seq([(c[2], c[1], 3, dict(stride=2, padding=1, indice_key='spconvd2', **dc))], norm_fn)
        self.comb_conv2 = _seq([(c[1] + c[1], c[1], 3, dict(padding=1, indice_key='submd2'))], norm_fn)
        self.conv11 = _seq([(c[0], c[0], 3, dict(stride=1, padding=1, indice_key='spconv11', **sp))], norm_fn)
        self.deconv1 = _seq([(c[1], c[1], 3, dict(stride=2, padding=1, indice_key='spconvd1', **dc))], norm_fn)
        self.comb_conv1 = _seq([(c[1] + c[0], c[1], 3, dict(padding=1, indice_key='submd1'))], norm_fn)
        self.num_point_features = c[1]


class VoxelBackBoneInverseRes(_ResDecoderBase):
    """the same topology on submanifold laterals with SparseInverseConv3d decoders that reuse the encoder's strided
    rulebooks ('spconv3', 'spconv2') and therefore restore the encoder's active sets exactly (spconv_backbone.py:385-527)"""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = grid_size[::-1]
        c = [16, 32, 64]
        sp, inv = dict(conv_type='spconv'), dict(conv_type='inverseconv')
        self.conv1 = _seq([(input_channels, c[0], 3, dict(padding=1, indice_key='subm1'))], norm_fn)
        self.conv2 = _seq([(c[0], c[1], 3, dict(stride=2, padding=1, indice_key='spconv2', **sp)),
                           (c[1], c[1], 3, dict(padding=1, indice_key='subm2'))], norm_fn)
        self.conv3 = _seq([(c[1], c[2], 3, dict(stride=2, padding=1, indice_key='spconv3', **sp)),
                           (c[2], c[2], 3, dict(padding=1, indice_key='subm3'))], norm_fn)
        self.conv22 = _seq([(c[1], c[1], 3, dict(stride=1, padding=1, indice_key='subm2'))], norm_fn)
        self.deconv2 = _seq([(c[2], c[1], 3, dict(stride=2, padding=1, indice_key='spconv3', **inv))], norm_fn)
        self.comb_conv2 = _seq([(c[1] + c[1], c[1], 3, dict(padding=1, indice_key='subm2'))], norm_fn)
        self.conv11 = _seq([(c[0], c[0], 3, dict(stride=1, padding=1, indice_key='subm1'))], norm_fn)
        self.deconv1 = _seq([(c[1], c[1], 3, dict(stride=2, padding=1, indice_key='spconv2', **inv))], norm_fn)
        self.comb_conv1 = _seq([(c[1] + c[0], c[1], 3, dict(padding=1, indice_key='subm1'))], norm_fn)
        self.num_point_features = c[1]


class VoxelResBackBone8x(nn.Module):
    """8x-downsampling detection backbone built from SparseBasicBlocks (spconv_backbone.py:531-627)"""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = grid_size[::-1] + [1, 0, 0]
        self.conv_input = spconv.SparseSequential(spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'),
                                                  norm_fn(16), nn.ReLU())
        res = lambda ch, key: SparseBasicBlock(ch, ch, norm_fn=norm_fn, indice_key=key)
        down = lambda ci, co, pad, key: post_act_block(ci, co, 3, norm_fn=norm_fn, stride=2, padding=pad, indice_key=key, conv_type='spconv')
        self.conv1 = spconv.SparseSequential(res(16, 'res1'), res(16, 'res1'))
        self.conv2 = spconv.SparseSequential(down(16, 32, 1, 'spconv2'), res(32, 'res2'), res(32, 'res2'))
        self.conv3 = spconv.SparseSequential(down(32, 64, 1, 'spconv3'), res(64, 'res3'), res(64, 'res3'))
        self.conv4 = spconv.SparseSequential(down(64, 128, (0, 1, 1), 'spconv4'), res(128, 'res4'), res(128, 'res4'))
        last_pad = self.model_cfg.get('last_pad', 0)
        self.conv_out = spconv.SparseSequential(spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                                                    indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128

    def forward(self, batch_dict):
        feats, coords = batch_dict['voxel_features'], batch_dict['voxel_coords'].int()
        x = spconv.SparseConvTensor(features=feats, indices=coords, spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        out = self.conv_out(x4)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8,
                           'multi_scale_3d_features': {'x_conv1': x1, 'x_conv2': x2, 'x_conv3': x3, 'x_conv4': x4}})
        return batch_dict


__all__ = {'VoxelBackBoneDeconv': VoxelBackBoneDeconv, 'VoxelBackBone8xOcc': VoxelBackBone8xOcc,
           'VoxelBackBoneDeconvRes': VoxelBackBoneDeconvRes, 'VoxelBackBoneInverseRes': VoxelBackBoneInverseRes,
           'VoxelResBackBone8x': VoxelResBackBone8x}
