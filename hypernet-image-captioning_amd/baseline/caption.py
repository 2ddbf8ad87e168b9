"""CATR captioner front and back ends of the reference's baseline/caption.py, with the hypernet-generated input
projection of BASELINE config 5.

  * `HyperInputProj` stands where `self.input_proj = nn.Conv2d(2048, hidden_dim, kernel_size=1)` stands
    (baseline/caption.py:13-14): the 1x1 convolution's weight [hidden, 2048] and bias are the output of a hypernetwork
    conditioned on a style / domain embedding.  The reference has NO implementation of this (cc_train_catr.py is a
    stub without a hypernet), so the design is this package's: the same hn_base + one-head-per-generated-tensor layout
    and M = 500 sizing rule as hypernet_attention.py:55-99, applied to the convolution's two parameters.  Parity is
    therefore UNPINNED by the reference; tests compare with torch's conv2d / linear on the same generated weights.
  * `MLP` is the 3-layer vocabulary head of baseline/caption.py:34-47.
  * `Caption` wires backbone -> projection -> transformer -> MLP as baseline/caption.py:9-32 does.  The transformer
    body (baseline/transformer.py: 6+6 pre-norm layers, multi-head attention, LayerNorm) is NOT built in this round:
    pass any module with the reference's call signature.

Every contraction here is a libcaphn kernel (caphn_hyper_*, caphn_gemm_f32, caphn_colsum_f32); there is no CPU path.
"""
import torch
from torch import nn

from caphn import functional as CF
from caphn import ops
from hypernet_attention import build_hypernet_layers


class HyperInputProj(nn.Module):
    def __init__(self, in_channels: int = 2048, hidden_dim: int = 256, hyper_emb: int = 10):
        super().__init__()
        self.in_channels, self.hidden_dim, self.hyper_emb = in_channels, hidden_dim, hyper_emb
        template = nn.Conv2d(in_channels, hidden_dim, kernel_size=1)          # only sizes the heads; not registered
        self.hn_base, self.hn_heads = build_hypernet_layers(template, hyper_emb)
        self._shape = ops.HyperShape(hyper_emb, [(h[0].out_features, h[2].out_features) for h in self.hn_heads])
        assert self._shape.theta_size == hidden_dim * in_channels + hidden_dim
        self.rank1_factors = None        # set by caphn.optim.FusedAdam.adopt_rank1(): dict filled by the backward

    def hyper_named_tensors(self):
        t = {"hn_base.0.weight": self.hn_base[0].weight, "hn_base.0.bias": self.hn_base[0].bias,
             "hn_base.2.weight": self.hn_base[2].weight, "hn_base.2.bias": self.hn_base[2].bias}
        for i, h in enumerate(self.hn_heads):
            t[f"hn_heads.{i}.0.weight"] = h[0].weight; t[f"hn_heads.{i}.0.bias"] = h[0].bias
            t[f"hn_heads.{i}.2.weight"] = h[2].weight; t[f"hn_heads.{i}.2.bias"] = h[2].bias
        return t

    def generate(self, x_style):
        """(weight [hidden, in_channels], bias [hidden]) for this style row; attached to the hypernet's graph."""
        theta = CF.hyper_forward(self._shape, x_style, self.hyper_named_tensors(), factor_sink=self.rank1_factors)
        n = self.hidden_dim * self.in_channels
        return theta[:n].view(self.hidden_dim, self.in_channels), theta[n:]

    def forward(self, src, x_style):
        """src [bs, in_channels, h, w] (backbone features, baseline/caption.py:24), x_style [hyper_emb] or [1, hyper_emb]
        -> [bs, hidden_dim, h, w], what nn.Conv2d(in_channels, hidden_dim, 1) returns for the generated weights."""
        w, b = self.generate(x_style)
        return CF.conv1x1(src, w, b)


class MLP(nn.Module):
    """baseline/caption.py:34-47: Linear+ReLU (num_layers - 1 times), then Linear."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        return CF.mlp(x, self.layers)


class Caption(nn.Module):
    """baseline/caption.py:9-32 with the generated projection.  backbone(samples) -> (features, pos) with features[-1]
    either an object with .decompose() (the reference's NestedTensor) or a (tensor, mask) pair."""

    def __init__(self, encodeur, transformer, hidden_dim, vocab_size, hyper_emb: int = 10, in_channels: int = 2048):
        super().__init__()
        self.backbone = encodeur
        self.input_proj = HyperInputProj(in_channels, hidden_dim, hyper_emb)
        self.transformer = transformer
        self.mlp = MLP(hidden_dim, 512, vocab_size, 3)

    def forward(self, samples, target, target_mask, x_style):
        features, pos = self.backbone(samples)
        last = features[-1]
        src, mask = last.decompose() if hasattr(last, "decompose") else last
        assert mask is not None
        hs = self.transformer(self.input_proj(src, x_style), mask, pos[-1], target, target_mask)
        return self.mlp(hs.permute(1, 0, 2))
