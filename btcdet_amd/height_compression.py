"""HeightCompression: sparse detection features -> dense BEV map (/root/reference/btcdet/models/backbones_2d/map_to_bev/
height_compression.py:10-26).  `.dense()` is the HIP scatter kernel btc_dense_fwd (its backward the gather btc_dense_bwd);
folding the depth axis into the channels is a view."""
from torch import nn


class HeightCompression(nn.Module):
    """keys read: encoded_spconv_tensor(+_stride); keys written: spatial_features (B, C*D, H, W), spatial_features_stride"""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        volume = batch_dict['encoded_spconv_tensor'].dense()          # (B, C, D, H, W), contiguous
        batch_dict.update(spatial_features=volume.flatten(1, 2),      # contiguous input: flatten is a view, like the reference's
                          spatial_features_stride=batch_dict['encoded_spconv_tensor_stride'])
        return batch_dict


__all__ = {'HeightCompression': HeightCompression}
