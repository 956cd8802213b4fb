"""Batch assembly of the hot path's inputs.

  * ``collate_batch(batch_list)``   -- drop-in for ``DatasetTemplate.collate_batch``
    (/root/reference/btcdet/datasets/dataset.py:167-223): per-scene numpy dicts -> one batch dict with the reference's keys,
    shapes and dtypes.  Checked key-for-key against the reference's own function on full-size scenes
    (tests/golden/gen_golden_full.py -> ``col_*`` vectors; tests/test_golden_full_cpu.py).
  * ``load_data_to_gpu(batch_dict)`` -- drop-in for ``btcdet.models.load_data_to_gpu`` (models/__init__.py:16-22): every
    ndarray becomes a float32 device tensor (integer coordinates and counts included -- the modules cast back with
    ``.int()``), the listed bookkeeping keys stay on the host.
  * ``collate_device(scenes)``      -- the resident counterpart: per-scene DEVICE tensors -> the same layout without a host
    round trip (the batch-index column is written by one kernel-free ``torch.cat`` of pre-built columns).

The rules are a table (key -> how scenes are joined) instead of the reference's if/elif chain; unknown keys are stacked, as
there."""
import numpy as np
import torch

# how a key's per-scene values are joined
CONCAT = "concat"            # rows of all scenes one after another
INDEXED = "indexed"          # same, with the scene index prepended as column 0
PAD_BOXES = "pad_boxes"      # (B, max_boxes, C) zero padded; also records gt_boxes_num
PAD_ROWS = "pad_rows"        # (B, max_rows, C) zero padded
PAD_FLAGS = "pad_flags"      # (B, max_boxes) float32 zero padded
AS_LIST = "list"             # left as a python list
RULES = {
    "voxels": CONCAT, "voxel_num_points": CONCAT, "voxel_points_label": CONCAT, "det_voxels": CONCAT, "det_voxel_num_points": CONCAT,
    "points": INDEXED, "voxel_coords": INDEXED, "det_voxel_coords": INDEXED, "bm_points": INDEXED,
    "gt_boxes": PAD_BOXES, "coverage_rates": PAD_ROWS, "box_mirr_flag": PAD_FLAGS,
    "miss_points": AS_LIST, "self_points": AS_LIST, "other_points": AS_LIST, "miss_occ_points": AS_LIST, "self_occ_points": AS_LIST,
    "other_occ_points": AS_LIST,
}
DROPPED = ("aug_boxes_image_idx", "aug_boxes_gt_idx", "aug_boxes_obj_ids", "obj_ids")
COUNTED = {"voxel_num_points": "batch_voxel_num", "det_voxel_num_points": "batch_det_voxel_num"}
HOST_KEYS = ("frame_id", "metadata", "calib", "image_shape", "timestamp_micros", "augment_box_num")


def _with_scene_column(rows, scene):
    rows = np.asarray(rows)
    col = np.full((rows.shape[0], 1), scene, dtype=rows.dtype)
    return np.concatenate([col, rows], axis=1)


def _pad_stack(values, width=None, dtype=np.float32):
    longest = max(len(v) for v in values)
    shape = (len(values), longest) if width is None else (len(values), longest, width)
    out = np.zeros(shape, dtype=dtype)
    for b, v in enumerate(values):
        if len(v):
            out[b, :len(v)] = np.asarray(v, dtype=dtype)
    return out


def collate_batch(batch_list, _unused=False):
    per_key = {}
    for sample in batch_list:
        for key, val in sample.items():
            per_key.setdefault(key, []).append(val)
            if key in COUNTED:
                per_key.setdefault(COUNTED[key], []).append(val.shape[0])
    for key in DROPPED:
        per_key.pop(key, None)
    out = {}
    for key, values in per_key.items():
        rule = RULES.get(key)
        try:
            if rule == CONCAT:
                out[key] = np.concatenate(values, axis=0)
            elif rule == INDEXED:
                out[key] = np.concatenate([_with_scene_column(v, b) for b, v in enumerate(values)], axis=0)
            elif rule == PAD_BOXES:
                out["gt_boxes_num"] = [len(v) for v in values]
                out[key] = _pad_stack(values, width=values[0].shape[-1])
            elif rule == PAD_ROWS:
                out[key] = _pad_stack(values, width=values[0].shape[-1])
            elif rule == PAD_FLAGS:
                out[key] = _pad_stack(values)
            elif rule == AS_LIST:
                out[key] = values
            else:
                out[key] = np.stack(values, axis=0)
        except Exception as e:  # the reference prints the key and raises TypeError
            raise TypeError("collate_batch: cannot join key %r (%s)" % (key, e))
    out["batch_size"] = len(batch_list)
    out["is_train"] = out["is_train"][0]
    return out


def load_data_to_gpu(batch_dict, device="cuda", keep_integers=False):
    """in place, like the reference.  keep_integers=True is this repository's leaner variant: integer arrays keep an integer
    device dtype (int32) instead of going through float32 -- the modules here accept both (they call ``.int()``)."""
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray) or key in HOST_KEYS:
            continue
        t = torch.from_numpy(val)
        if keep_integers and not t.dtype.is_floating_point and t.dtype != torch.bool:
            batch_dict[key] = t.to(device=device, dtype=torch.int32)
        else:
            batch_dict[key] = t.float().to(device)
    return batch_dict


def collate_device(scenes, is_train=True):
    """scenes: list of dicts of DEVICE tensors with the per-scene keys of DataProcessor.forward (voxels, voxel_coords (M,3),
    voxel_num_points, det_*, points (N,4), bm_points (Nb,3), gt_boxes (G,8), box_mirr_flag (G), rot_z scalar) -> the batch
    layout of collate_batch + load_data_to_gpu (float32 everywhere), built on the device"""
    B = len(scenes)
    dev = scenes[0]["points"].device
    out = {"batch_size": B, "is_train": is_train}
    for key in ("voxels", "voxel_num_points", "det_voxels", "det_voxel_num_points"):
        if key in scenes[0]:
            out[key] = torch.cat([s[key] for s in scenes], dim=0).float()
            if key in COUNTED:
                out[COUNTED[key]] = torch.tensor([s[key].shape[0] for s in scenes], dtype=torch.float32, device=dev)
    for key in ("points", "voxel_coords", "det_voxel_coords", "bm_points"):
        if key in scenes[0]:
            parts = [torch.cat([torch.full((s[key].shape[0], 1), float(b), dtype=torch.float32, device=dev), s[key].float()], dim=1)
                     for b, s in enumerate(scenes)]
            out[key] = torch.cat(parts, dim=0)
    counts = [int(s["gt_boxes"].shape[0]) for s in scenes]
    G = max(counts)
    gt = torch.zeros((B, G, scenes[0]["gt_boxes"].shape[-1]), dtype=torch.float32, device=dev)
    flags = torch.zeros((B, G), dtype=torch.float32, device=dev)
    for b, s in enumerate(scenes):
        if counts[b]:
            gt[b, :counts[b]] = s["gt_boxes"].float()
            if "box_mirr_flag" in s:
                flags[b, :counts[b]] = s["box_mirr_flag"].float()
    out["gt_boxes"], out["gt_boxes_num"], out["box_mirr_flag"] = gt, counts, flags
    if "rot_z" in scenes[0]:
        out["rot_z"] = torch.stack([torch.as_tensor(s["rot_z"], dtype=torch.float32, device=dev).reshape(()) for s in scenes])
    return out
