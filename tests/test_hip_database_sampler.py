"""btcdet_amd.database_sampler.DataBaseSampler against the REFERENCE's own class (tests/golden/gen_sampler_golden.py ->
sampler.npz): same database, same three consecutive scenes, same global-RNG seed -> identical pasted boxes, names, index
bookkeeping and point clouds.  The BEV-IoU rejection test runs through the HIP kernel (btc_boxes_pairwise_bev); only its
ZERO / non-zero outcome matters here, so the comparison is exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import common  # noqa: E402

pytestmark = pytest.mark.gpu


class ED(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


def scenes():
    from btcdet_amd import synth
    out = []
    for seed in (31, 32, 33):
        s = synth.make_scene(seed, az_step=0.8)
        n = s["gt_boxes"].shape[0]
        out.append({"points": s["points"].copy(), "gt_boxes": s["gt_boxes"][:, :7].copy(), "gt_names": np.array(["Car"] * n),
                    "gt_boxes_mask": np.array([True] * n), "gt_boxes_inds": np.arange(n)})
    if len(out[1]["gt_boxes"]) > 1:
        out[1]["gt_boxes_mask"][0] = False
    return out


@pytest.mark.parametrize("variant,extra", [("plain", [0.0, 0.0, 0.0]), ("wide_limit", [0.2, 0.2, 0.2])])
def test_database_sampler_vs_reference(tmp_path, variant, extra):
    from btcdet_amd.database_sampler import DataBaseSampler
    g = np.load(os.path.join(HERE, "golden", "sampler.npz"))
    cfg = ED(PREPARE={"filter_by_min_points": ["Car:5", "Pedestrian:5"], "filter_by_difficulty": [-1]}, SAMPLE_GROUPS=["Car:15", "Pedestrian:4"],
             NUM_POINT_FEATURES=4, DATABASE_WITH_FAKELIDAR=False, REMOVE_EXTRA_WIDTH=extra, LIMIT_WHOLE_SCENE=variant == "wide_limit",
             USE_ROAD_PLANE=False)
    infos = common.make_gt_database(tmp_path)
    sampler = DataBaseSampler(tmp_path, cfg, ["Car", "Pedestrian"], infos)
    np.random.seed(99)
    for i, sc in enumerate(scenes()):
        r = sampler(sc)
        p = "%s%d_" % (variant, i)
        assert "gt_boxes_mask" not in r
        np.testing.assert_array_equal(r["gt_boxes"], g[p + "gt_boxes"])
        assert [str(x) for x in r["gt_names"]] == [str(x) for x in g[p + "gt_names"]]
        np.testing.assert_array_equal(r["gt_boxes_inds"], g[p + "gt_boxes_inds"])
        assert int(r.get("augment_box_num", 0)) == int(g[p + "augment_box_num"]) > 0
        np.testing.assert_array_equal(r["aug_boxes_image_idx"], g[p + "aug_boxes_image_idx"])
        np.testing.assert_array_equal(r["aug_boxes_gt_idx"], g[p + "aug_boxes_gt_idx"])
        np.testing.assert_array_equal(r["points"], g[p + "points"])
