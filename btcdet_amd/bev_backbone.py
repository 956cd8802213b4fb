"""2-D BEV backbone between HeightCompression and the anchor head (SURVEY.md §8f row 1 "glue"): dense 3x3 conv / BatchNorm /
ReLU pyramids whose levels are brought back to one resolution by (transposed) convs and concatenated.  The convolutions are the
vendor library's (MIOpen through torch.nn) -- dense 2-D convs are not this repository's kernels; what matters here is the
module protocol and parameter names (`blocks.N.M`, `deblocks.N.M`), which equal the reference's
(/root/reference/btcdet/models/backbones_2d/base_bev_backbone.py:6-112), so its checkpoints load.
Keys read: spatial_features; written: spatial_features_2d."""
import numpy as np
import torch
from torch import nn


def _cbr(conv):
    return [conv, nn.BatchNorm2d(conv.out_channels, eps=1e-3, momentum=0.01), nn.ReLU()]


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        depths = list(model_cfg.get("LAYER_NUMS", None) or [])
        strides = list(model_cfg.get("LAYER_STRIDES", None) or [])
        widths = list(model_cfg.get("NUM_FILTERS", None) or [])
        up_strides = list(model_cfg.get("UPSAMPLE_STRIDES", None) or [])
        up_widths = list(model_cfg.get("NUM_UPSAMPLE_FILTERS", None) or [])
        assert len(depths) == len(strides) == len(widths) and len(up_strides) == len(up_widths)
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        cin = input_channels
        for lvl, (depth, stride, width) in enumerate(zip(depths, strides, widths)):
            layers = [nn.ZeroPad2d(1)] + _cbr(nn.Conv2d(cin, width, 3, stride=stride, padding=0, bias=False))
            for _ in range(depth):
                layers += _cbr(nn.Conv2d(width, width, 3, padding=1, bias=False))
            self.blocks.append(nn.Sequential(*layers))
            if up_strides:
                s = up_strides[lvl]
                if s >= 1:
                    up = nn.ConvTranspose2d(width, up_widths[lvl], s, stride=s, bias=False)
                else:   # a fractional factor means a strided conv down to the common resolution
                    k = int(np.round(1 / s))
                    up = nn.Conv2d(width, up_widths[lvl], k, stride=k, bias=False)
                self.deblocks.append(nn.Sequential(up, nn.BatchNorm2d(up_widths[lvl], eps=1e-3, momentum=0.01), nn.ReLU()))
            cin = width
        self.num_bev_features = sum(up_widths)
        if len(up_strides) > len(depths):   # one more transposed conv over the concatenation
            c, s = self.num_bev_features, up_strides[-1]
            self.deblocks.append(nn.Sequential(nn.ConvTranspose2d(c, c, s, stride=s, bias=False), nn.BatchNorm2d(c, eps=1e-3, momentum=0.01), nn.ReLU()))

    def forward(self, data_dict):
        x = data_dict["spatial_features"]
        full = x.shape[2]
        merged = []
        for lvl, block in enumerate(self.blocks):
            x = block(x)
            data_dict["spatial_features_%dx" % int(full / x.shape[2])] = x
            merged.append(self.deblocks[lvl](x) if len(self.deblocks) > 0 else x)
        if merged:
            x = torch.cat(merged, dim=1) if len(merged) > 1 else merged[0]
        if len(self.deblocks) > len(self.blocks):
            x = self.deblocks[-1](x)
        data_dict["spatial_features_2d"] = x
        return data_dict


__all__ = {"BaseBEVBackbone": BaseBEVBackbone}
