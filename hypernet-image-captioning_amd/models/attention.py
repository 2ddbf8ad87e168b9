"""BahdanauAttention with the reference's constructor, sub-module names and return tuple
(models/attention.py:5-46).  Inside AttentionGru's loops the arithmetic is fused into libcaphn's
recurrent kernels; this stand-alone forward serves callers that step a decoder by hand (beam search,
hypernet_attention.py:264-272) and runs on libcaphn too: two GEMMs and one fused score / softmax /
context kernel, differentiable (caphn.functional.bahdanau_attention)."""
import torch
from torch import nn
from torch.nn import functional as F

from caphn import functional as CF


class BahdanauAttention(nn.Module):
    def __init__(self, num_features, hidden_dim, output_dim=1):
        super().__init__()
        self.num_features = num_features
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.W_a = nn.Linear(self.num_features, self.hidden_dim)
        self.U_a = nn.Linear(self.hidden_dim, self.hidden_dim)
        self.v_a = nn.Linear(self.hidden_dim, self.output_dim)

    def forward(self, features, decoder_hidden):
        """features [B,P,F], decoder_hidden [B,H] -> (context [B,F], atten_weight [B,P])."""
        if self.output_dim != 1:
            raise NotImplementedError("BahdanauAttention with output_dim != 1 (the reference only builds 1, models/attention.py:9)")
        return CF.bahdanau_attention(features, decoder_hidden, self.W_a, self.U_a, self.v_a)
