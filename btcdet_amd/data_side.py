"""Data-side producers of the hot path's inputs: what the reference's dataset code hands to the model
besides the raw scan -- `pre_rot_points` / `rot_z` (global rotation augmentation with SAVE_PRE_ROT), the scaled / flipped
special point sets, `bm_points` (best-match templates placed into the ground-truth boxes) -- and the two on-disk formats
behind them.  Host-side mirrors with the reference's names, argument meaning, RNG draws and in-place behaviour
(tests/test_data_side_cpu.py checks them against vectors produced by the reference's own functions,
tests/golden/gen_data_side_golden.py), plus device-resident batched forms for a pipeline that keeps scenes in HBM
(bench.py's synthetic scenes carry these keys already; btcdet_amd/synth.py).

  read_kitti_bin / write_kitti_bin     KittiDataset.get_lidar, kitti_dataset.py:72-75: float32 (N, 4) [x, y, z, intensity]
  read_bm_template                     multi_best_match_querier.py:63-67: pickle of a flat float array -> (-1, F)[:, :3]
  rotate_points_along_z                common_utils.py:34-56 (fp32 torch matmul with [[c, s, 0], [-s, c, 0], [0, 0, 1]])
  global_rotation / global_scaling / random_flip_along_x     augmentor_utils.py:5-20,44-82
  random_world_rotation                data_augmentor.py:136-155 (adds pre_rot_points and rot_z in DEGREES)
  best_match_points                    MltBestMatchQuerier.__call__, multi_best_match_querier.py:50-76,269-296
  rotate_scenes_on_device              the rotation of a resident stacked batch (rot_z per scene), one launch"""
import pickle

import numpy as np
import torch

SPECIAL_NAMES = ["bm_points", "miss_points", "self_points", "other_points", "miss_occ_points", "self_occ_points", "other_occ_points",
                 "self_limit_occ_mask", "miss_full_occ_points", "other_full_occ_points"]  # data_augmentor.py:8


def read_kitti_bin(path):
    return np.fromfile(str(path), dtype=np.float32).reshape(-1, 4)


def write_kitti_bin(path, points):
    np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4).tofile(str(path))


def read_bm_template(path, load_point_features=3):
    with open(str(path), "rb") as f:
        obj_points = pickle.load(f)
    return obj_points.reshape([-1, load_point_features])[:, :3].astype(np.float32)


def rotate_points_along_z(points, angle):
    """points (B, N, 3 + C), angle (B) radians, x towards y; numpy in -> numpy out (computed in fp32 torch, as the reference)"""
    is_numpy = isinstance(points, np.ndarray)
    pts = torch.from_numpy(points).float() if is_numpy else points
    ang = torch.from_numpy(angle).float() if isinstance(angle, np.ndarray) else angle
    cosa, sina = torch.cos(ang), torch.sin(ang)
    zeros, ones = ang.new_zeros(ang.shape[0]), ang.new_ones(ang.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    out = torch.cat((torch.matmul(pts[:, :, 0:3], rot), pts[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def global_rotation(gt_boxes, points, rot_range, special_points_lst=()):
    """one np.random.uniform draw; gt_boxes is modified in place and returned, like the reference"""
    special_points_lst = list(special_points_lst)
    noise_rotation = np.random.uniform(rot_range[0], rot_range[1])
    ang = np.array([noise_rotation])
    points = rotate_points_along_z(points[np.newaxis, :, :], ang)[0]
    gt_boxes[:, 0:3] = rotate_points_along_z(gt_boxes[np.newaxis, :, 0:3], ang)[0]
    gt_boxes[:, 6] += noise_rotation
    if gt_boxes.shape[1] > 7:
        v = np.hstack((gt_boxes[:, 7:9], np.zeros((gt_boxes.shape[0], 1))))[np.newaxis, :, :]
        gt_boxes[:, 7:9] = rotate_points_along_z(v, ang)[0][:, 0:2]
    for i in range(len(special_points_lst)):
        special_points_lst[i] = rotate_points_along_z(special_points_lst[i][np.newaxis, :, :], ang)[0]
    return gt_boxes, points, noise_rotation, special_points_lst


def global_scaling(gt_boxes, points, scale_range, special_points_lst=()):
    special_points_lst = list(special_points_lst)
    if scale_range[1] - scale_range[0] < 1e-3:
        return gt_boxes, points, special_points_lst
    noise_scale = np.random.uniform(scale_range[0], scale_range[1])
    points[:, :3] *= noise_scale
    gt_boxes[:, :6] *= noise_scale
    for sp in special_points_lst:
        sp[:, :3] *= noise_scale
    return gt_boxes, points, special_points_lst


def random_flip_along_x(gt_boxes, points, special_points_lst=(), enable=None):
    special_points_lst = list(special_points_lst)
    enable = np.random.choice([False, True], replace=False, p=[0.5, 0.5]) if enable is None else enable
    if enable:
        gt_boxes[:, 1] = -gt_boxes[:, 1]
        gt_boxes[:, 6] = -gt_boxes[:, 6]
        points[:, 1] = -points[:, 1]
        if gt_boxes.shape[1] > 7:
            gt_boxes[:, 8] = -gt_boxes[:, 8]
        for sp in special_points_lst:
            sp[:, 1] = -sp[:, 1]
    return gt_boxes, points, special_points_lst


def random_world_rotation(data_dict, config):
    """config: WORLD_ROT_ANGLE (scalar or [lo, hi], radians), SAVE_PRE_ROT -> pre_rot_points + rot_z (degrees)"""
    rot_range = config['WORLD_ROT_ANGLE']
    if not isinstance(rot_range, list):
        rot_range = [-rot_range, rot_range]
    pre_rot_points = data_dict['points']
    names = [k for k in SPECIAL_NAMES if k in data_dict]
    gt_boxes, points, noise_rotation, special = global_rotation(data_dict['gt_boxes'], pre_rot_points, rot_range, [data_dict[k] for k in names])
    for k, v in zip(names, special):
        data_dict[k] = v
    data_dict['gt_boxes'] = gt_boxes
    data_dict['points'] = points
    if config.get("SAVE_PRE_ROT", False):
        data_dict['pre_rot_points'] = pre_rot_points
        data_dict['rot_z'] = noise_rotation * 180 / np.pi
    return data_dict


def get_yaw_rotation(yaw):
    """point_box_utils.py:50-59"""
    c, s = np.cos(yaw), np.sin(yaw)
    one, zero = np.ones_like(yaw), np.zeros_like(yaw)
    return np.stack([np.stack([c, -1.0 * s, zero], axis=-1), np.stack([s, c, zero], axis=-1), np.stack([zero, zero, one], axis=-1)], axis=-2)


def best_match_points(gt_boxes, gt_names, gt_boxes_inds, frame_id, template_root, class_names, load_point_features=3):
    """bm_points (sum n_i, 3): for every ground-truth box of an in-scope class, its best-match template
    `<root[class]>/<int(frame_id)>_<box id>.pkl` rotated by the box heading and moved to the box centre; boxes in input order"""
    image_idx = int(frame_id)
    lst = []
    for idx in range(len(gt_boxes_inds)):
        gt_box, gt_name = gt_boxes[idx], gt_names[idx]
        if gt_name in class_names:
            obj_points = read_bm_template(template_root[gt_name] / "{}_{}.pkl".format(image_idx, gt_boxes_inds[idx]), load_point_features)
            obj_points = np.einsum("nj,ij->ni", obj_points, get_yaw_rotation(gt_box[6])) + gt_box[:3]
            lst.append(obj_points)
    if len(lst) > 1:
        return np.concatenate(lst, axis=0)[..., :3]
    if len(lst) == 1:
        return lst[0][..., :3]
    return np.zeros([0, 3], dtype=np.float32)


def rotate_scenes_on_device(points, scene_offsets, rot_z_deg):
    """resident form of the SAVE_PRE_ROT rotation: points (sum N, 3 + C) CUDA, scene_offsets (B + 1) int, rot_z_deg (B) degrees
    -> rotated copy (the input is the `pre_rot_points` of the batch).  Same arithmetic per point as rotate_points_along_z
    (fp32: x' = x c - y s, y' = x s + y c)."""
    counts = (scene_offsets[1:] - scene_offsets[:-1]).long()
    ang = torch.repeat_interleave(rot_z_deg.float() * (np.pi / 180.0), counts)
    c, s = torch.cos(ang), torch.sin(ang)
    x, y = points[:, 0], points[:, 1]
    out = points.clone()
    out[:, 0] = x * c + y * (-s)
    out[:, 1] = x * s + y * c
    return out
