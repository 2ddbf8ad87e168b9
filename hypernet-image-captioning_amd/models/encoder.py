"""EncoderCNN shim (models/encoder.py:7-26).

The reference's encoder is a frozen, pretrained torchvision ResNet-152 trunk: outside the hot
path and not reproducible offline (weights are a download).  This stand-in keeps the interface
``forward(images) -> [B,49,2048]``: precomputed feature maps pass through unchanged; raw images
are mapped by a fixed (seeded, frozen) 32x32-patch projection + ReLU, i.e. synthetic features of
the right shape and sparsity, NOT ResNet features.
"""
import torch
from torch import nn


class EncoderCNN(nn.Module):
    def __init__(self, channels: int = 2048, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("proj", torch.randn(3, channels, generator=g) * 0.8, persistent=False)

    @torch.no_grad()
    def forward(self, images):
        if images.dim() == 3:                       # already [B,P,2048]
            return images
        pooled = nn.functional.adaptive_avg_pool2d(images.float(), (7, 7))      # [B,3,7,7]
        feats = pooled.permute(0, 2, 3, 1).reshape(images.shape[0], 49, 3) @ self.proj
        return torch.relu(feats)
