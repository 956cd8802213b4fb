"""The hot path assembled as the reference assembles it: BtcNet's occupancy branch and the sparse part of its
detection branch, with the module lists, registry names and ``occ_modules`` / ``det_modules`` containers of
/root/reference/btcdet/models/detectors/detector3d_template.py:28-113 and btcnet.py:32-56, so that
``state_dict`` keys line up with reference checkpoints (occ_modules.backbone_3d.conv1.0.0.weight, ...).

The hot path proper ends at HeightCompression; a training step drives its backward with a stand-in L2
loss on the two tensors the heads behind it consume (``spatial_features`` and ``multi_scale_3d_features['x_combine']``,
btcdet_amd/trainer.py).  ``heads="rpn"`` appends the §8f row-1 glue -- BaseBEVBackbone + AnchorHeadSingle with the
reference's module names (``det_modules.backbone_2d`` / ``det_modules.dense_head``) and its RPN loss (btcnet.py:108-114);
the ROI head (ConvHead) is not built, ``x_combine`` keeps its stand-in.
"""
import numpy as np
import torch
import torch.nn as nn

from . import backbones_3d, bev_backbone, dense_head, height_compression, occ_head, occ_targets, pass_occ_vox, vfe
from .processor import DataProcessor


def _device_tensors(obj, depth=0):
    """the device tensors inside a batch_dict value: a tensor (anything with record_stream), or a container (tuple / list / dict / an
    object with __slots__ such as spconv.ops.Rulebook) of them"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
    elif hasattr(obj, "record_stream"):
        yield obj
    elif depth < 5 and not isinstance(obj, (str, bytes, int, float, bool, type(None))):
        if isinstance(obj, dict):
            items = list(obj.values())
        elif isinstance(obj, (tuple, list)):
            items = obj
        elif hasattr(obj, "__slots__"):
            items = [getattr(obj, a, None) for a in obj.__slots__]
        else:
            return
        for v in items:
            yield from _device_tensors(v, depth + 1)


DET_GEOMETRY_AHEAD = True   # forward_occ walks the detection backbone's rulebooks right behind PassOccVox (False: forward_det does, as before round 6)


class HotPathDataset(object):
    """the attributes Detector3DTemplate reads from its dataset (dataset.py:27-41)"""

    def __init__(self, cfg, training=True):
        d = cfg.DATA_CONFIG
        self.dataset_cfg = d
        self.class_names = cfg.CLASS_NAMES
        self.training = training
        self.mode = 'train' if training else 'test'
        self.point_cloud_range = np.array(d.POINT_CLOUD_RANGE, dtype=np.float32)
        self.occ_point_cloud_range = np.array(d.OCC.POINT_CLOUD_RANGE, dtype=np.float32)
        self.num_point_features = len(d.POINT_FEATURE_ENCODING.used_feature_list)
        self.data_processor = DataProcessor(d.DATA_PROCESSOR, point_cloud_range=self.occ_point_cloud_range, training=training,
                                            occ_config=d.OCC, det_point_cloud_range=self.point_cloud_range)
        self.occ_grid_size, self.occ_voxel_size = self.data_processor.occ_grid_size, self.data_processor.occ_voxel_size
        self.det_grid_size, self.det_voxel_size = self.data_processor.det_grid_size, self.data_processor.det_voxel_size
        self.occ_dim = self.data_processor.occ_dim


class BtcHotPath(nn.Module):
    def __init__(self, cfg, dataset=None, device="cuda", heads=None):
        super().__init__()
        assert heads in (None, "rpn", "full"), heads
        self.cfg = cfg
        self.heads = heads
        self.dataset = dataset if dataset is not None else HotPathDataset(cfg)
        ds, m, d = self.dataset, cfg.MODEL, cfg.DATA_CONFIG
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.voxel_centers = occ_targets.cylinder_voxel_centers(ds.occ_grid_size, d.OCC.POINT_CLOUD_RANGE, d.OCC.VOXEL_SIZE, device)
        nraw = ds.num_point_features
        self.occ_modules, self.det_modules = nn.Module(), nn.Module()
        occ_num_class = 1 if m.OCC.PARAMS.CLASS_AGNOSTIC else len(cfg.CLASS_NAMES)
        # ---- occupancy branch (occ_module_topology, detector3d_template.py:32-34)
        tgt = occ_targets.__all__[m.OCC.TARGETS.NAME](model_cfg=m.OCC, point_cloud_range=ds.occ_point_cloud_range,
                                                       voxel_size=ds.occ_voxel_size, data_cfg=d, grid_size=ds.occ_grid_size,
                                                       num_class=occ_num_class, voxel_centers=self.voxel_centers)
        ovfe = vfe.__all__[m.OCC.VFE.NAME](model_cfg=m.OCC.VFE, num_point_features=nraw, point_cloud_range=ds.point_cloud_range,
                                           voxel_size=ds.occ_voxel_size, data_cfg=d, grid_size=ds.occ_grid_size,
                                           num_class=occ_num_class, maxprob=False)
        obb = backbones_3d.__all__[m.OCC.BACKBONE_3D.NAME](model_cfg=m.OCC.BACKBONE_3D, input_channels=ovfe.get_output_feature_dim(),
                                                           grid_size=ds.occ_grid_size, voxel_size=ds.occ_voxel_size,
                                                           point_cloud_range=ds.point_cloud_range,
                                                           original_num_rawpoint_features=nraw)
        head = occ_head.__all__[m.OCC.OCC_DENSE_HEAD.NAME](model_cfg=m.OCC, data_cfg=d, input_channels=obb.num_point_features,
                                                           num_class=occ_num_class, grid_size=ds.occ_grid_size)
        upd = pass_occ_vox.__all__[m.OCC.OCC_PNT_UPDATE.NAME](model_cfg=m.OCC, data_cfg=d, point_cloud_range=ds.point_cloud_range,
                                                              occ_voxel_size=ds.occ_voxel_size, occ_grid_size=ds.occ_grid_size,
                                                              det_voxel_size=ds.det_voxel_size, det_grid_size=ds.det_grid_size,
                                                              mode=ds.mode, voxel_centers=self.voxel_centers)
        for name, mod in [("occ_targets", tgt), ("vfe", ovfe), ("backbone_3d", obb), ("occ_dense_head", head), ("occ_pnt_update", upd)]:
            self.occ_modules.add_module(name, mod)
        self.occ_module_list = [tgt, ovfe, obb, head, upd]
        # ---- detection branch up to the BEV map (module_topology, detector3d_template.py:28-30)
        nvox = nraw + upd.code_num_dim
        dvfe = vfe.__all__[m.VFE.NAME](model_cfg=m.VFE, num_point_features=nvox, point_cloud_range=ds.point_cloud_range,
                                       voxel_size=ds.det_voxel_size, data_cfg=d, grid_size=ds.det_grid_size,
                                       num_class=len(cfg.CLASS_NAMES), maxprob=d.OCC.get("MAX_VFE", False))
        dbb = backbones_3d.__all__[m.BACKBONE_3D.NAME](model_cfg=m.BACKBONE_3D, input_channels=dvfe.get_output_feature_dim(),
                                                       grid_size=ds.det_grid_size, voxel_size=ds.det_voxel_size,
                                                       point_cloud_range=ds.point_cloud_range, original_num_rawpoint_features=nraw)
        bev = height_compression.__all__[m.MAP_TO_BEV.NAME](model_cfg=m.MAP_TO_BEV, grid_size=ds.det_grid_size, occ_dim=ds.occ_dim)
        for name, mod in [("vfe", dvfe), ("backbone_3d", dbb), ("map_to_bev_module", bev)]:
            self.det_modules.add_module(name, mod)
        self.det_module_list = [dvfe, dbb, bev]
        if heads in ("rpn", "full"):   # module_topology continues: backbone_2d, dense_head [, roi_head] (detector3d_template.py:28-30,252-330)
            b2d = bev_backbone.__all__[m.BACKBONE_2D.NAME](model_cfg=m.BACKBONE_2D, input_channels=m.MAP_TO_BEV.NUM_BEV_FEATURES)
            dh = dense_head.__all__[m.DENSE_HEAD.NAME](model_cfg=m.DENSE_HEAD, input_channels=b2d.num_bev_features,
                                                       num_class=len(cfg.CLASS_NAMES) if not m.DENSE_HEAD.CLASS_AGNOSTIC else 1,
                                                       class_names=cfg.CLASS_NAMES, grid_size=ds.det_grid_size,
                                                       point_cloud_range=ds.point_cloud_range, predict_boxes_when_training=heads == "full")
            self.det_modules.add_module("backbone_2d", b2d)
            self.det_modules.add_module("dense_head", dh)
            self.det_module_list += [b2d, dh]
            if heads == "full":   # the ROI head behind the proposals (btcnet.py:116-121; conv_head.py) -- training targets: roi_targets.py
                from . import conv_head
                rh = conv_head.__all__[m.ROI_HEAD.NAME](input_channels=dbb.num_point_features, model_cfg=m.ROI_HEAD,
                                                        num_class=len(cfg.CLASS_NAMES) if not m.ROI_HEAD.CLASS_AGNOSTIC else 1,
                                                        det_voxel_size=ds.det_voxel_size, point_cloud_range=ds.point_cloud_range,
                                                        num_rawpoint_features=nraw)
                self.det_modules.add_module("roi_head", rh)
                self.det_module_list.append(rh)
        self.percentage = d.OCC.get("USEOCC_PERCENTAGE", 1.0)

    # ------------------------------------------------------------------------------------------------------------------
    # The weight-independent front of a step.  In the reference the voxelizations run in DataLoader workers (dataset.py:
    # 185-192, DataProcessor.transform_points_to_voxels) and overlap the GPU step for free; here they are GPU work, and so are
    # the occupancy targets (functions of the voxels and the boxes) and every rulebook of the occupancy branch (functions of
    # the voxel coordinates).  prepare() runs all of that for one batch; given a side stream it runs THERE, so a training loop
    # calls it for batch k+1 right after enqueueing batch k's backward: the read-backs inside (voxel counts, rulebook row
    # counts) then wait for the side stream only, not for the backward pass still running on the main stream, and the next
    # forward finds its first ~1.5 ms of host work free of synchronisation points.
    # ------------------------------------------------------------------------------------------------------------------
    def assemble(self, batch, is_train=True):
        """raw scene batch (synth.make_batch keys on the device) -> batch_dict after both voxelizations"""
        proc = self.dataset.data_processor
        bd = proc.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"])
        bd.update({"batch_size": batch["batch_size"], "points": batch["points5"], "gt_boxes": batch["gt_boxes"],
                   "gt_boxes_num": batch["gt_boxes_num"], "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"],
                   "rot_z": batch["rot_z"], "is_train": is_train})
        return bd

    def _prepare(self, batch, is_train):
        bd = self.assemble(batch, is_train)
        n_done = 0
        for mod in self.occ_module_list[:2]:  # occupancy targets, parameter-free VFE
            if any(p.requires_grad for p in mod.parameters()):
                break
            bd = mod(bd)
            n_done += 1
        bb = self.occ_modules.backbone_3d
        if n_done == 2 and hasattr(bb, "prefetch_geometry"):
            bd = bb.prefetch_geometry(bd, self.occ_modules.occ_dense_head)
        if self.heads in ("rpn", "full") and bd["is_train"]:
            # anchor targets are a function of the boxes alone (dense_head.AxisAlignedTargetAssigner: nonzero / argmax with host
            # read-backs, as in the reference): part of the weight-independent front, off the training thread's stream
            dh = self.det_modules.dense_head
            bd["rpn_targets"] = dh.target_assigner.assign_targets(dh.anchors(bd["gt_boxes"].device), bd["gt_boxes"])
        bd["__prepared__"] = n_done
        return bd

    def prepare(self, batch, stream=None, is_train=True):
        """-> batch_dict for forward().  stream: a torch.cuda.Stream to run on (see above); may be called from a worker thread
        while the main thread sits in loss.backward().  The tensors produced on `stream` live in that stream's allocator pool:
        a generation of them is kept alive here until the step that consumed it has ended on the main stream
        (mark_step_end()) AND the side stream has been made to wait for that point -- only then may the pool hand its blocks
        to a later prepare().  The batch's own tensors must be complete when this is called."""
        if stream is None or not batch["points"].is_cuda:
            return self._prepare(batch, is_train)
        st = self.__dict__.setdefault("_prep_state", {"gens": [], "next": 0, "consumed": -1})
        gens = st["gens"]
        if st["next"] == 0:
            stream.wait_stream(torch.cuda.current_stream())  # whatever produced the first batch
        keep = []
        for g in gens:
            if g["ended"] is not None:      # consumed, its step has ended: order the side stream behind that, then release
                stream.wait_event(g["ended"])
            elif len(gens) - len(keep) > 3:  # nobody calls mark_step_end(): fall back to "behind everything enqueued so far"
                stream.wait_stream(torch.cuda.current_stream())
            else:
                keep.append(g)
        gens[:] = keep
        with torch.cuda.stream(stream):
            bd = self._prepare(batch, is_train)
            ready = torch.cuda.Event()
            ready.record(stream)
        bd["__ready_event__"] = ready
        bd["__generation__"] = st["next"]
        gens.append({"id": st["next"], "refs": list(bd.values()), "ended": None})
        st["next"] += 1
        return bd

    def mark_step_end(self, stream=None, upto=None):
        """call once per training step after the backward pass (and optimizer) have been enqueued: the generations of
        prepared tensors consumed so far may be recycled once `stream` (default: the current stream) has passed this point.
        upto: the generation of the batch whose step is ending (batch_dict["__gen_id__"]) -- a loop that runs the occupancy
        branch of the NEXT batch before this step has ended (bench.make_step, pipelined) must not release that one yet."""
        st = self.__dict__.get("_prep_state")
        if not st:
            return
        limit = st["consumed"] if upto is None else min(st["consumed"], upto)
        ev = None
        for g in st["gens"] + self.__dict__.get("_borrowed", []):   # (_borrowed: forward_det's share, see _borrow)
            if g["ended"] is None and g["id"] <= limit:
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(stream if stream is not None else torch.cuda.current_stream())
                g["ended"] = ev

    def forward_occ(self, batch_dict):
        """the occupancy branch of BtcNet.forward (btcnet.py:32-44) and its loss (btcnet.py:91-101):
        -> (batch_dict, occ_loss, tb_dict, event recorded when the detection branch's inputs are complete).  The detection
        branch is detached from this one (PASS_GRAD False), so a training loop may run loss.backward() of this branch
        while forward_det() runs on another stream (bench.make_step)."""
        ready = batch_dict.pop("__ready_event__", None)
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        n_done = batch_dict.pop("__prepared__", 0)
        gen = batch_dict.pop("__generation__", None)
        if gen is not None:
            self._prep_state["consumed"] = max(self._prep_state["consumed"], gen)
            batch_dict["__gen_id__"] = gen
        use_occ_prob = [True] * batch_dict["batch_size"]
        prob = np.random.uniform(size=batch_dict["batch_size"], high=0.9999)  # the reference consumes this stream too
        if batch_dict["is_train"]:
            use_occ_prob = prob <= self.percentage
        batch_dict["use_occ_prob"] = use_occ_prob
        head = self.occ_modules.occ_dense_head
        if hasattr(head, "premerge"):
            head.premerge()  # the merged head weight is built before the backbone runs, so its CatBackward runs after it
        prepared = {id(v) for v in batch_dict.values()} if gen is not None else None
        for mod in self.occ_module_list[n_done:]:
            batch_dict = mod(batch_dict)
        dbb = self.det_modules.backbone_3d
        if DET_GEOMETRY_AHEAD and hasattr(dbb, "prefetch_geometry") and "voxel_coords" in batch_dict and batch_dict["voxel_coords"].is_cuda:
            # the detection backbone's rulebooks are a function of the coordinates PassOccVox has just made: walked here, on the thread
            # that has them first (see VoxelBackBone8xOcc.prefetch_geometry)
            batch_dict = dbb.prefetch_geometry(batch_dict)
        if prepared is not None:
            # what THIS call produced (on the current stream's pool): the only tensors a consumer on another stream has to register
            # (hand_over) -- the prepared front is kept alive by its generation until the step has ended on every stream
            batch_dict["__produced_here__"] = [v for v in batch_dict.values() if torch.is_tensor(v) and v.is_cuda and id(v) not in prepared]
            if "det_geometry" in batch_dict:     # (the rulebooks' tensors live in this stream's pool too: kept alive with the rest)
                batch_dict["__produced_here__"].append(batch_dict["det_geometry"])
        det_inputs_ready = None
        if torch.cuda.is_available() and batch_dict["voxels"].is_cuda:
            det_inputs_ready = torch.cuda.Event()
            det_inputs_ready.record()
        occ_loss, tb_dict = head.get_loss(batch_dict)
        return batch_dict, occ_loss, tb_dict, det_inputs_ready

    def _borrow(self, batch_dict):
        """what hand_over() achieves with ~40 record_stream calls (0.25 ms of the training thread per step, tools/host_sampler.py), by
        holding on instead: the tensors the occupancy branch produced on ITS stream stay referenced here until the consuming stream
        has passed the end of the step that read them (mark_step_end() attaches that event; the references go once the event has
        COMPLETED -- a host-side query, no stream is made to wait).  Their blocks cannot return to the producer's pool earlier, which
        is all record_stream would have ensured.  Only for batches of a loop that calls mark_step_end() (a prepared batch: __gen_id__)."""
        pend = self.__dict__.setdefault("_borrowed", [])
        # Steps end in order on the consuming stream, and every forward_det contains a BLOCKING read-back on that stream (the rulebook
        # walk's level sizes, or PassOccVox's counts on the producer's side): when this is called for generation g, the last such wait
        # -- in step g - 1 -- was enqueued behind the end-of-step event of generation g - 2, which has therefore completed.  So
        # generations up to g - 3 are dropped without asking the driver (an event query is ~50 us: the queries were 0.24 ms of the
        # training thread per step, on the thread whose host time is the step); anything older than six generations is queried.
        gen = batch_dict["__gen_id__"]
        while pend and pend[0]["ended"] is not None and pend[0]["id"] <= gen - 3:
            pend.pop(0)
        while len(pend) >= 6 and pend[0]["ended"] is not None and pend[0]["ended"].query():
            pend.pop(0)
        while sum(b["ended"] is None for b in pend) > 3:   # nobody calls mark_step_end(): register the oldest with the consumer after all
            old = next(b for b in pend if b["ended"] is None)
            pend.remove(old)
            for t in _device_tensors(old["refs"]):
                t.record_stream(torch.cuda.current_stream())
        pend.append({"id": batch_dict["__gen_id__"], "refs": list(batch_dict.pop("__produced_here__")), "ended": None})

    @staticmethod
    def hand_over(batch_dict, stream):
        """register every device tensor of batch_dict with `stream`, which will consume them:
        they were allocated on the producer's stream, whose pool would otherwise hand their blocks out again while `stream` still
        reads them.  ~40 calls of ~10 us: the pipelined step makes them from the producer's thread, off the training thread."""
        for t in _device_tensors(batch_dict.pop("__produced_here__", None) or list(batch_dict.values())):
            t.record_stream(stream)
        batch_dict["__recorded_for__"] = stream.cuda_stream

    def forward_det(self, batch_dict, inputs_ready=None):
        """the detection branch up to the BEV map (btcnet.py:46-56) on the CURRENT stream; inputs_ready: the event of
        forward_occ when that ran on another stream (the tensors it produced are then also registered with this stream)"""
        if inputs_ready is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(inputs_ready)
            if batch_dict.pop("__recorded_for__", None) != cur.cuda_stream:   # (hand_over() did it from the producer's thread)
                if batch_dict.get("__gen_id__") is not None and batch_dict.get("__produced_here__") is not None:
                    self._borrow(batch_dict)
                else:
                    self.hand_over(batch_dict, cur)
        rows = {}   # active rows per level of this batch -- tensor shapes, host integers, no read-back (bench.py reports them per timed step)
        enc = batch_dict.get("encoded_spconv_tensor")
        if enc is not None and hasattr(enc, "features"):
            rows["occ_out"] = int(enc.features.shape[0])          # the occupancy decoder's output level (the head runs on it)
        if "det_voxel_coords" in batch_dict:
            rows["det_voxels"] = int(batch_dict["det_voxel_coords"].shape[0])
        rows["det_voxels_after_pass_occ"] = int(batch_dict["voxel_coords"].shape[0])   # M'': detection voxels + PassOccVox's added ones
        for mod in self.det_module_list:
            batch_dict = mod(batch_dict)
        rows.update(batch_dict.get("__level_rows__", {}))
        self.last_level_rows = rows
        # the two tensors the heads behind the hot path consume: the BEV map (BaseBEVBackbone) and x_combine (ConvHead)
        return {"spatial_features": batch_dict["spatial_features"],
                "x_combine": batch_dict["multi_scale_3d_features"]["x_combine"].features}, batch_dict

    def det_loss(self, ret, batch_dict):
        """the detection branch's training loss (btcnet.py:108-129): with heads="rpn" the anchor head's RPN loss + the L2
        stand-in for the ROI head's consumer tensor; without heads the two stand-ins of trainer.stand_in_det_loss"""
        from .trainer import MeanSquare, stand_in_det_loss
        if self.heads not in ("rpn", "full"):
            return stand_in_det_loss(ret, batch_dict)
        loss_rpn, tb = self.det_modules.dense_head.get_loss()
        if self.heads == "full":    # get_training_loss: loss_rpn + loss_rcnn (btcnet.py:108-121)
            loss_rcnn, tb = self.det_modules.roi_head.get_loss(tb)
            return loss_rpn + loss_rcnn
        return loss_rpn + MeanSquare.apply(ret["x_combine"], 1e-3)

    def forward(self, batch_dict):
        """BtcNet.forward up to the BEV map (btcnet.py:32-56) + the occupancy loss (btcnet.py:91-101), in order on the current stream --
        what a caller of the plain module protocol gets (the reference's own train_one_epoch_multi_opt: model(batch) -> loss.backward()
        -> optimizer steps).  (The detection branch on a stream of its own inside this call measured slower, 284 against 326 scenes/s:
        one more active stream fed by the same host thread costs more than the overlap buys; HotPathTrainer's schedules give each
        stream its own thread.)"""
        batch_dict, occ_loss, tb_dict, _ = self.forward_occ(batch_dict)
        out, batch_dict = self.forward_det(batch_dict)
        out["loss_occ"] = occ_loss
        return out, tb_dict, batch_dict
