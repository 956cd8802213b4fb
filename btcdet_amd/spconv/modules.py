"""SparseModule / SparseSequential (spconv v1.2.1 semantics):
sparse modules receive the SparseConvTensor; plain nn.Modules (BatchNorm1d, ReLU) are applied to
``.features`` in place when the input is sparse and has at least one row.
Call sites: /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:34-42,106-128,656-700."""
from collections import OrderedDict

from torch import nn

from .tensor import SparseConvTensor


import torch as _torch

nn_float32 = _torch.float32
FUSE_CONV_BN = True  # hand the BatchNorm (+ReLU) that follows a sparse conv to the conv's autograd node
CHAIN_LAYERS = True  # a container of conv -> BN -> ReLU layers with ready rulebooks runs as ONE compiled call (_chain_plan)
FUSE_BN_RELU = True  # set False to run BatchNorm1d / ReLU through torch (used by the parity test)


class SparseModule(nn.Module):
    """marker base class of modules that take a SparseConvTensor"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward_geometry(self, input):
        """the sparse layers' rulebooks only (SparseConvolution.forward_geometry); dense modules are skipped"""
        for module in self._modules.values():
            if is_spconv_module(module):
                input = module.forward_geometry(input)
        return input

    def _flat_modules(self):
        """the leaf modules in execution order, nested plain SparseSequentials (post_act_blocks inside a stage) expanded"""
        out = []
        for m in self._modules.values():
            if type(m) is SparseSequential:
                out.extend(m._flat_modules())
            else:
                out.append(m)
        return out

    def _chain_plan(self, input):
        """[(conv, bn, relu, rulebook, inverse)] when this container is nothing but conv -> BatchNorm1d (-> ReLU) layers whose
        rulebooks ALL exist already in input.indice_dict (by indice_key or in the geometry cache -- the occupancy branch after
        BtcHotPath.prepare), else None.  No rulebook is built and no dict is written here."""
        from . import fused_bn, ops
        from .conv import SparseConvolution
        f = input.features
        if not (FUSE_CONV_BN and FUSE_BN_RELU and CHAIN_LAYERS and f is not None and f.is_cuda and f.shape[0] > 0
                and f.dtype in (nn_float32, _torch.bfloat16)):
            return None
        # the container's structure is fixed: its (conv, bn, relu) triples are derived once (None: not a pure conv -> BatchNorm -> ReLU chain)
        triples = self.__dict__.get("_chain_triples", False)
        if triples is False:
            mods = self._flat_modules()
            triples, i = [], 0
            while i < len(mods):
                conv = mods[i]
                bn = mods[i + 1] if i + 1 < len(mods) else None
                # structure only (a BatchNorm1d behind a 3-D sparse conv): whether the BatchNorm is fusable IN ITS CURRENT STATE is asked per
                # call below -- an eval sanity pass before training must not switch the fast path off for good
                if not (isinstance(conv, SparseConvolution) and not conv.conv1x1 and conv.ndim == 3 and isinstance(bn, nn.BatchNorm1d)) \
                        or ops.pads_in_channels(conv.in_channels):   # (padded per layer: ops._pad_in_channels)
                    triples = None
                    break
                relu = i + 2 < len(mods) and type(mods[i + 2]) is nn.ReLU
                triples.append((conv, bn, relu))
                i += 2 + int(relu)
            self.__dict__["_chain_triples"] = triples
        if not triples:
            return None
        plan = []
        indices, shape = input.indices, input.spatial_shape
        geom = input.indice_dict.get("__geometry_cache__", None)
        for conv, bn, relu in triples:
            if not fused_bn.fusable(bn):    # (a BatchNorm switched to eval without running statistics, ...)
                return None
            if f.dtype == _torch.bfloat16 and (conv.in_channels % 16 or conv.out_channels % 16):
                return None
            rb = input.indice_dict.get(conv.indice_key, None) if conv.indice_key is not None else None
            if conv.inverse:
                if rb is None or rb.n_in == 0:
                    return None
                indices, shape = rb.in_indices, rb.in_shape[3 - conv.ndim:]
            else:
                if rb is None:
                    hit = geom.get(conv._gkey(indices, shape), None) if geom is not None else None
                    rb = hit[0] if hit is not None else None
                if rb is None or isinstance(rb, ops.PendingRulebook) or rb.n_out == 0:
                    return None
                indices, shape = rb.out_indices, conv._out_shape(shape)
            plan.append((conv, bn, relu, rb, conv.inverse))
        return (plan, indices, shape) if plan else None

    def _run_chain(self, input, plan, indices, shape):
        from . import fused_bn, ops
        F = ops.fast()
        # what does not change from call to call (parameters, buffers, BatchNorm constants, workspace sizes) is gathered once per
        # (training flags, weight tensors) of the chain; per call only the rulebook's maps / row orders and the row counts differ
        # (nn.Module._apply -- .to(device), .float() -- and load_state_dict(assign=True) REPLACE buffer tensors and keep Parameter
        # identity: the buffers' and the affine parameters' identities are part of the key, or the chain would go on reading and
        # updating tensors the module no longer owns)
        key = tuple((bn.training, bn.track_running_stats, id(conv.weight), id(conv.bias), id(bn.weight), id(bn.bias), id(bn.running_mean),
                     id(bn.running_var), id(bn.num_batches_tracked), conv.weight.device) for conv, bn, _, _, _ in plan)
        st = self.__dict__.get("_chain_static")
        if st is None or st[0] != key:
            w, b, ga, be, rms, rvs, nbts, ub, mom, eps, relus, need, dfr = ([] for _ in range(13))
            for conv, bn, relu, rb, inverse in plan:
                training = bn.training or not bn.track_running_stats
                rm = bn.running_mean if bn.track_running_stats else None
                w.append(ops._f32c(conv.weight)); b.append(conv.bias)
                ga.append(bn.weight); be.append(bn.bias); rms.append(rm); rvs.append(bn.running_var if bn.track_running_stats else None)
                nbts.append(bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None)
                ub.append(bool(training or rm is None)); mom.append(float(bn.momentum)); eps.append(float(bn.eps)); relus.append(bool(relu))
                need.append(int(fused_bn._ws(input.features.device, conv.weight.shape[-1])[1])); dfr.append(bool(conv.weight.is_leaf))
            st = self.__dict__["_chain_static"] = (key, (w, b, ga, be, rms, rvs, nbts, ub, mom, eps, relus, need, dfr))
        w, b, ga, be, rms, rvs, nbts, ub, mom, eps, relus, need, dfr = st[1]
        mf, mb, of, ob, ov = [], [], [], [], []
        for conv, bn, relu, rb, inverse in plan:
            if inverse:
                mf.append(rb.nbr_in); mb.append(rb.nbr_out); of.append(rb.order_in); ob.append(rb.order_out)
            else:
                mf.append(rb.nbr_out); mb.append(rb.map_bwd); of.append(rb.order_out); ob.append(rb.order_in)
            ov.append(bool(ops._overlap_ok(mf[-1].shape[0])))
        ws = fused_bn._ws(input.features.device, max(wt.shape[-1] for wt in w))[0]
        out = F.conv_bn_relu_chain(ops._actc(input.features), w, b, mf, mb, of, ob, ga, be, rms, rvs, nbts, ub, mom, eps, relus, ws, need, ov, dfr)
        out_tensor = SparseConvTensor(out, indices, shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor

    def forward(self, input):
        from . import fused_bn, ops
        from .conv import SparseConvolution
        if isinstance(input, SparseConvTensor) and ops.PROFILE is None and ops.CAPTURE is None and ops.NATIVE_AUTOGRAD and ops.fast() is not None:
            plan = self._chain_plan(input)
            if plan is not None:
                return self._run_chain(input, *plan)
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                nxt = mods[i] if i < len(mods) else None
                f = input.features
                if FUSE_CONV_BN and FUSE_BN_RELU and nxt is not None and isinstance(module, SparseConvolution) and fused_bn.fusable(nxt) and f.is_cuda \
                        and f.dtype in (nn_float32, _torch.bfloat16) and f.shape[0] != 0:
                    # conv -> BatchNorm1d (-> ReLU): one autograd node, the same kernels (ops.SparseConvBNReLUFunction)
                    relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                    input = module(input, fuse_bn=(nxt, relu))
                    i += 1 + int(relu)
                else:
                    input = module(input)
            else:
                if isinstance(input, SparseConvTensor):
                    f = input.features
                    if fused_bn.is_sync(module) and f is not None and f.is_cuda and f.dtype in (nn_float32, _torch.bfloat16) and f.dim() == 2:
                        # --sync_bn: batch statistics over all ranks' rows (fused_bn.SyncBatchNormReLUFunction).  BEFORE the empty-tensor
                        # test: a rank without rows at this layer must still enter the all_gather / all_reduce the other ranks enter
                        relu = i < len(mods) and type(mods[i]) is nn.ReLU
                        input.features = fused_bn.sync_batch_norm_relu(module, f, relu)
                        i += int(relu)
                    elif input.indices.shape[0] != 0:
                        if FUSE_BN_RELU and fused_bn.fusable(module) and f.is_cuda and f.dtype in (nn_float32, _torch.bfloat16) and f.dim() == 2:
                            # BatchNorm1d (+ the ReLU that follows it): one fused HIP call, same parameters / buffers
                            relu = i < len(mods) and type(mods[i]) is nn.ReLU
                            input.features = fused_bn.batch_norm_relu(module, f, relu)
                            i += int(relu)
                        else:
                            input.features = module(f)
                    elif isinstance(module, nn.SyncBatchNorm) and module.training and f is not None and f.is_cuda:
                        # a torch SyncBatchNorm (what convert_sync_batchnorm puts in place of a layer it cannot mark: affine=False,
                        # momentum=None) on a rank WITHOUT rows at this layer: the other ranks enter its collectives, so this one must
                        # too (torch's SyncBatchNorm takes a 0-row batch); skipping it hangs the job
                        input.features = module(f)
                else:
                    input = module(input)
        return input
