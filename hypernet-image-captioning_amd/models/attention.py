"""BahdanauAttention with the reference's constructor, sub-module names and return tuple
(models/attention.py:5-46).  Inside AttentionGru's teacher-forced loop the arithmetic is fused
into libcaphn's persistent recurrent kernel; this stand-alone forward serves callers that step
the decoder by hand (beam search, hypernet_attention.py:264-272) and composes libcaphn GEMMs
with torch pointwise ops."""
import torch
from torch import nn
from torch.nn import functional as F

from caphn import ops


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
        if features.is_cuda and not torch.is_grad_enabled():
            B, P, Fd = features.shape
            f2 = features.reshape(B * P, Fd).contiguous().float()
            atten_1 = ops.gemm(f2, self.W_a.weight, False, True, bias=self.W_a.bias).view(B, P, -1)
            atten_2 = ops.gemm(decoder_hidden.contiguous().float(), self.U_a.weight, False, True,
                               bias=self.U_a.bias).unsqueeze(1)
        else:
            atten_1 = self.W_a(features)
            atten_2 = self.U_a(decoder_hidden.unsqueeze(1))
        atten_score = self.v_a(torch.tanh(atten_1 + atten_2))
        atten_weight = F.softmax(atten_score, dim=1)
        context = torch.sum(atten_weight * features, dim=1)
        return context, atten_weight.squeeze(dim=2)
