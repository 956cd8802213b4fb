"""The optimizer half of the data-parallel step, as the reference runs it per parameter group
(`occ_modules`, `det_modules`) every iteration -- /root/reference/tools/train_utils/train_utils.py:121-124:

    clip_grad_norm_(group parameters, GRAD_NORM_CLIP)            -> one L2 norm over the group, grads *= min(1, c / (norm + 1e-6))
    optimizer.step()            adam_onecycle = fastai OptimWrapper(Adam(betas=(mom, 0.99)), true_wd=True, bn_wd=True)
                                (optimization/__init__.py:29-44, fastai_optim.py:132-150): DECOUPLED weight decay
                                p *= 1 - wd * lr on every trainable parameter (BatchNorm included), then Adam with wd = 0
    lr_scheduler.step(it)       OneCycle (learning_schedules_fastai.py:64-81): lr and beta1 for the NEXT iteration, two cosine
                                phases around PCT_START
    optimizer.lr = max(lr, LR_CLIP)

Here: one multi-tensor norm, one multi-tensor scale of the parameters and one fused multi-tensor Adam launch per group
(torch._fused_adam_; the clip coefficient rides in as its grad_scale, so the gradients are not rewritten separately), no
Optimizer-wrapper Python.  Checked against the reference's own OptimWrapper + OneCycle + clip_grad_norm_ run in this container
(tests/golden/gen_optim_golden.py -> tests/golden/optim.npz, tests/test_train_step_cpu.py)."""
import math

import os

import torch


class OneCycle(object):
    """lr / beta1 the reference's OneCycle leaves in the optimizer after `step(it)`; `initial()` is what its constructor sets"""

    def __init__(self, total_step, lr_max, moms, div_factor, pct_start):
        self.total = int(total_step)
        self.turn = int(self.total * pct_start)
        self.lr_max = float(lr_max)
        self.lr_low = self.lr_max / float(div_factor)
        self.lr_end = self.lr_low / 1e4
        self.mom_hi, self.mom_lo = float(moms[0]), float(moms[1])

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    def initial(self):
        return self.lr_low, self.mom_hi

    def at(self, it):
        if it < self.turn:
            pct = it / float(self.turn)
            return self._cos(self.lr_low, self.lr_max, pct), self._cos(self.mom_hi, self.mom_lo, pct)
        pct = (min(it, self.total) - self.turn) / float(self.total - self.turn)
        return self._cos(self.lr_max, self.lr_end, pct), self._cos(self.mom_lo, self.mom_hi, pct)


_BN_TYPES = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm)


def reference_param_groups(module):
    """the two torch param groups the reference's OptimWrapper.create builds for a module (optimization/__init__.py:29-44 ->
    fastai_optim.split_bn_bias): trainable parameters of the non-BatchNorm leaf modules in traversal order, then those of the
    BatchNorm leaves.  Their concatenation is the parameter numbering of its optimizer state_dict."""
    leaves = [m for m in module.modules() if len(list(m.children())) == 0]
    plain = [p for m in leaves if not isinstance(m, _BN_TYPES) for p in m.parameters() if p.requires_grad]
    norm = [p for m in leaves if isinstance(m, _BN_TYPES) for p in m.parameters() if p.requires_grad]
    return plain, norm


def chunk_tables(sizes, device):
    """chunk tables of csrc/optim.hip for parameters of the given sizes laid out back to back in a flat buffer: chunks of <= 1024
    elements that never straddle two parameters -> dict(seg, off, len, flat: device tensors; seg0: host list, first chunk of each
    parameter)"""
    from . import _lib
    max_seg = int(_lib.lib().btc_adam_max_segments())   # pointer-table width of the loaded library (csrc/optim.hip GradPtrs)
    off, seg, coff, clen, cflat, seg0 = 0, [], [], [], [], [0]
    for i, n in enumerate(sizes):
        for c0 in range(0, n, 1024):
            seg.append(i % max_seg); coff.append(c0); clen.append(min(1024, n - c0)); cflat.append(off + c0)
        seg0.append(len(seg))
        off += n
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    return dict(seg=i32(seg), off=i32(coff), len=i32(clen), flat=torch.tensor(cflat, dtype=torch.int64, device=device), seg0=seg0)


class GroupOptimizer(object):
    """one or more parameter groups, each {"params" or "module", "lr" (LR of the yaml = lr_max), "weight_decay",
    "grad_norm_clip", "moms", "div_factor", "pct_start", "lr_clip"}; total_steps = iterations per epoch x epochs.
    With "module" the group's parameters are taken (and numbered, for state_dict_lst) as the reference's optimizer does."""

    def __init__(self, groups, total_steps, beta2=0.99, eps=1e-8, flat=None):
        """flat (default: on for CUDA parameters when the compiled binding is there): a group's
        parameters and Adam moments become views into one flat buffer each and the whole step of the group is three launches of
        csrc/optim.hip instead of ~10 multi-tensor torch ops over ~120-tensor lists (same arithmetic; the host side of those ops
        was what the GPU waited for).  A step in which some parameter has no (or an unusual) gradient takes the list path."""
        self.groups = []
        for g in groups:
            if "module" in g:
                plain, norm = reference_param_groups(g["module"])
                g = dict(g, params=plain + norm, split=len(plain))
            params = [p for p in g["params"] if p.requires_grad]
            sched = OneCycle(total_steps, g["lr"], g.get("moms", (0.95, 0.85)), g.get("div_factor", 10.0), g.get("pct_start", 0.4))
            lr, mom = sched.initial()
            self.groups.append(dict(params=params, sched=sched, lr=lr, mom=mom, weight_decay=float(g.get("weight_decay", 0.0)),
                                    clip=float(g.get("grad_norm_clip", 0.0)), lr_clip=float(g.get("lr_clip", 1e-7)), split=g.get("split", len(params)),
                                    exp_avgs=[torch.zeros_like(p) for p in params], exp_avg_sqs=[torch.zeros_like(p) for p in params],
                                    steps=[torch.zeros((), dtype=torch.float32, device=p.device) for p in params], it=0))
        self.beta2, self.eps = beta2, eps
        self._present = self._missing = None
        if flat is None:
            flat = True
        if flat:
            for g in self.groups:
                self._make_flat(g)

    def _make_flat(self, g):
        from . import _lib
        params = g["params"]
        F = _lib.fast() if (params and all(p.is_cuda and p.dtype == torch.float32 for p in params)) else None
        if F is None or len({p.device for p in params}) != 1:
            return
        dev = params[0].device
        sizes = [p.numel() for p in params]
        total = sum(sizes)
        with torch.no_grad():
            flat_p = torch.empty((total,), dtype=torch.float32, device=dev)
            flat_m = torch.zeros((total,), dtype=torch.float32, device=dev)
            flat_v = torch.zeros((total,), dtype=torch.float32, device=dev)
            off = 0
            for i, (p, n) in enumerate(zip(params, sizes)):
                flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat_p[off:off + n].view(p.shape)           # the module's parameter now lives in the flat buffer
                flat_m[off:off + n].copy_(g["exp_avgs"][i].reshape(-1))
                flat_v[off:off + n].copy_(g["exp_avg_sqs"][i].reshape(-1))
                g["exp_avgs"][i] = flat_m[off:off + n].view(p.shape)
                g["exp_avg_sqs"][i] = flat_v[off:off + n].view(p.shape)
                off += n
            t = chunk_tables(sizes, dev)
            ws_bytes = _lib.lib().btc_adam_group_ws_bytes(int(t["seg"].numel()))
            g["flat"] = dict(F=F, p=flat_p, m=flat_m, v=flat_v, seg=t["seg"], off=t["off"], len=t["len"], flat=t["flat"], seg0=t["seg0"],
                             ws=torch.zeros((ws_bytes,), dtype=torch.uint8, device=dev), n=0, synced=True)

    def _sync_steps(self, g):
        """per-parameter step counters (list path, checkpoints) from the flat path's single count"""
        fl = g.get("flat")
        if fl is not None and not fl["synced"]:
            for st in g["steps"]:
                st.fill_(float(fl["n"]))
            fl["synced"] = True

    def _flat_step(self, g, grads):
        """the group's step through csrc/optim.hip; False if this step has to take the list path"""
        fl = g.get("flat")
        if fl is None or fl["n"] < 0:
            return False
        for gr in grads:
            if gr is None or gr.dtype != torch.float32 or not gr.is_contiguous():
                self._sync_steps(g)
                fl["n"] = -1        # per-parameter step counts may differ from here on: list path for good
                return False
        from ._lib import stream_ptr
        fl["n"] += 1
        fl["synced"] = False
        fl["F"].adam_group_step(g["params"], grads, fl["seg"], fl["off"], fl["len"], fl["flat"], fl["seg0"], fl["p"], fl["m"], fl["v"], fl["n"],
                                float(g["lr"]), float(g["mom"]), float(self.beta2), float(self.eps), float(g["weight_decay"]), float(g["clip"]),
                                fl["ws"], stream_ptr())
        return True

    @property
    def iteration(self):
        """optimizer steps taken (the OneCycle position; groups stepped separately within a training step count once)"""
        return max(g["it"] for g in self.groups) if self.groups else 0

    @iteration.setter
    def iteration(self, value):
        for g in self.groups:
            g["it"] = int(value)

    def read_grads_from(self, view_of, present=None, missing=None):
        """take the gradients from fixed buffers (a gradient reducer's flat buckets) instead of param.grad.  present(param):
        whether the parameter received a gradient this step -- one that did not is skipped, as torch.optim.Adam skips
        grad-is-None parameters (its bucket slice holds zeros, which Adam would otherwise treat as a real gradient);
        missing(): the set of ids of such parameters (normally empty: the per-parameter queries are then skipped)"""
        for g in self.groups:
            g["grad_views"] = [view_of(p) for p in g["params"]]
        self._present, self._missing = present, missing

    def zero_grad(self, set_to_none=True):
        for g in self.groups:
            pl = self._param_list(g)
            if pl is not None:
                g["flat"]["F"].clear_grads_pl(pl)      # one call instead of ~300 attribute stores (0.13 ms of Python per step)
                continue
            for p in g["params"]:
                p.grad = None

    @staticmethod
    def _param_list(g):
        """the group's parameters as an object of the compiled binding (binding.cpp ParamList: built once), or None"""
        fl = g.get("flat")
        if fl is None or not hasattr(fl.get("F"), "make_param_list"):
            return None
        pl = fl.get("plist")
        if pl is None:
            pl = fl["plist"] = fl["F"].make_param_list(list(g["params"]))
        return pl

    def _flat_step_params(self, g):
        """_flat_step with the gradients read from the parameters' .grad inside the binding (no 300-element Python lists per step);
        False: a gradient is missing or unusual, nothing was launched -- the caller builds the lists and decides"""
        fl = g.get("flat")
        if fl is None or fl["n"] < 0:
            return False
        pl = self._param_list(g)
        if pl is None:
            return False
        from ._lib import stream_ptr
        if not fl["F"].adam_group_step_pl(pl, fl["seg"], fl["off"], fl["len"], fl["flat"], fl["seg0"], fl["p"], fl["m"], fl["v"], fl["n"] + 1,
                                          float(g["lr"]), float(g["mom"]), float(self.beta2), float(self.eps), float(g["weight_decay"]),
                                          float(g["clip"]), fl["ws"], stream_ptr()):
            return False
        fl["n"] += 1
        fl["synced"] = False
        return True

    def lrs(self):
        return [g["lr"] for g in self.groups]

    @torch.no_grad()
    def step(self, groups=None):
        """groups: indices of the parameter groups to step (default: all).  The groups are independent -- own gradient-norm
        clip, own schedule, as the reference's two optimizers are -- so a training loop may step a group as soon as ITS
        gradients are complete (bench.py steps the detection branch's group while the occupancy branch is still in backward)."""
        last_norms = []   # per clipped group: the gradient norm (list path) / a view of the SQUARED norm, a double (flat path)
        for gi, g in enumerate(self.groups):
            if groups is not None and gi not in groups:
                continue
            params = g["params"]
            if params and "grad_views" not in g and self._flat_step_params(g):
                if g["clip"] > 0:
                    last_norms.append(g["flat"]["ws"][:8].view(torch.float64))
                nlr, nmom = g["sched"].at(g["it"])
                g["lr"], g["mom"] = max(nlr, g["lr_clip"]), nmom
                g["it"] += 1
                continue
            if "grad_views" in g:
                grads = g["grad_views"]
                if self._present is None or (self._missing is not None and not self._missing()):
                    keep = params   # all present (only its length is used below)
                else:
                    keep = [i for i, p in enumerate(params) if self._present(p)]
            else:
                grads = [p.grad for p in params]
                keep = [i for i, gr in enumerate(grads) if gr is not None]
            # (a missing gradient retires the flat path of the group: per-parameter step counts differ from then on)
            if params and "flat" in g and self._flat_step(g, grads if len(keep) == len(params) else [None]):
                if g["clip"] > 0:   # the squared norm stays on the device (first 8 bytes of the workspace); no launch for logging
                    last_norms.append(g["flat"]["ws"][:8].view(torch.float64))
                nlr, nmom = g["sched"].at(g["it"])
                g["lr"], g["mom"] = max(nlr, g["lr_clip"]), nmom
                g["it"] += 1
                continue
            if len(keep) != len(params):
                params, grads = [params[i] for i in keep], [grads[i] for i in keep]
                ea, es, st = [g["exp_avgs"][i] for i in keep], [g["exp_avg_sqs"][i] for i in keep], [g["steps"][i] for i in keep]
            else:
                ea, es, st = g["exp_avgs"], g["exp_avg_sqs"], g["steps"]
            if not params:   # nothing to update (no gradient reached the group): the schedule still advances
                nlr, nmom = g["sched"].at(g["it"])
                g["lr"], g["mom"] = max(nlr, g["lr_clip"]), nmom
                g["it"] += 1
                continue
            inv_coef = None
            if g["clip"] > 0:
                total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
                # grads *= min(1, clip / (norm + 1e-6))  ==  grads /= max(1, (norm + 1e-6) / clip): handed to the fused kernel
                inv_coef = ((total + 1e-6) / g["clip"]).clamp_(min=1.0).to(torch.float32)
                last_norms.append(total)
            lr = g["lr"]
            if g["weight_decay"] != 0.0:
                torch._foreach_mul_(params, 1.0 - g["weight_decay"] * lr)
            torch._foreach_add_(st, 1)
            torch._fused_adam_(params, grads, ea, es, [], st, lr=lr, beta1=g["mom"], beta2=self.beta2, weight_decay=0.0, eps=self.eps,
                               amsgrad=False, maximize=False, grad_scale=inv_coef, found_inf=None)
            nlr, nmom = g["sched"].at(g["it"])
            g["lr"], g["mom"] = max(nlr, g["lr_clip"]), nmom
            g["it"] += 1
        return last_norms

    # ---- checkpoint format of the reference: one torch.optim.Adam state_dict per optimizer (train_utils.py:272-288) -------------
    def state_dict_lst(self):
        """[state_dict per group] in torch.optim.Adam's layout with the reference's two param groups (non-BatchNorm, BatchNorm)
        and its parameter numbering, so that a checkpoint written here resumes in the reference and vice versa"""
        out = []
        for g in self.groups:
            self._sync_steps(g)
            n, split = len(g["params"]), g["split"]
            state = {}
            for i in range(n):
                if float(g["steps"][i]) > 0:
                    state[i] = {"step": g["steps"][i].detach().clone(), "exp_avg": g["exp_avgs"][i].detach().clone(),
                                "exp_avg_sq": g["exp_avg_sqs"][i].detach().clone()}
            common = dict(lr=g["lr"], betas=(g["mom"], self.beta2), eps=self.eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                          capturable=False, differentiable=False, fused=None)
            pgs = [dict(common, params=list(range(0, split))), dict(common, params=list(range(split, n)))]
            out.append({"state": state, "param_groups": pgs})
        return out

    def load_state_dict_lst(self, states, iteration=None):
        """inverse of state_dict_lst; iteration: the `it` of the checkpoint (the OneCycle position; lr / beta1 are also restored
        from the param groups, which is what the reference's resume does)"""
        assert len(states) == len(self.groups)
        for g, sd in zip(self.groups, states):
            n = len(g["params"])
            assert sum(len(pg["params"]) for pg in sd["param_groups"]) == n, "optimizer state does not match the parameter count"
            for i in range(n):
                st = sd["state"].get(i)
                if st is None:
                    g["steps"][i].zero_(); g["exp_avgs"][i].zero_(); g["exp_avg_sqs"][i].zero_()
                else:
                    g["steps"][i].fill_(float(st["step"]))
                    g["exp_avgs"][i].copy_(st["exp_avg"])
                    g["exp_avg_sqs"][i].copy_(st["exp_avg_sq"])
            pg = sd["param_groups"][-1]
            g["lr"], g["mom"] = float(pg["lr"]), float(pg["betas"][0])
            if "flat" in g:   # one step count for the whole group, or the list path
                counts = {int(float(st)) for st in g["steps"]}
                g["flat"]["n"] = counts.pop() if len(counts) == 1 else -1
                g["flat"]["synced"] = True
        if iteration is not None:
            self.iteration = int(iteration)
