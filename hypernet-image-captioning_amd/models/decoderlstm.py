"""Decoder models with the reference's class names, constructor signatures, sub-module names
and return types (models/decoderlstm.py:11-185).  Sub-modules stay ordinary nn.Modules holding
ordinary nn.Parameters in the reference's layout, so callers can call, swap and checkpoint them
(hypernet_attention.py:251-272, :419-428); the teacher-forced forward reads their storage and
runs entirely in libcaphn's HIP kernels."""
import torch
from torch import nn

from caphn import functional as CF
from caphn import ops
from caphn._lib import CaphnError
from .attention import BahdanauAttention


class AttentionGru(nn.Module):
    def __init__(self, num_features, feature_out, embedding_dim, hidden_dim, vocab_size, num_layers=1, p=0.0):
        super().__init__()
        self.num_features = num_features
        self.feature_out = feature_out
        self.embedding_dim = embedding_dim
        self.hidden_dim = hidden_dim
        self.vocab_size = vocab_size
        self.num_layers = num_layers
        self.sample_temp = 0.5
        self.feature_fc = nn.Sequential(nn.Linear(num_features, feature_out), nn.ReLU(),
                                        nn.Linear(feature_out, feature_out))
        self.embed = nn.Embedding(vocab_size, embedding_dim)
        self.gru = nn.GRUCell(embedding_dim + feature_out, hidden_dim)
        self.layers = None
        if num_layers > 1:
            self.layers = nn.ModuleList([nn.GRUCell(input_size=hidden_dim, hidden_size=hidden_dim)
                                         for _ in range(num_layers - 1)])
        self.fc = nn.Linear(hidden_dim, vocab_size)
        self.attention = BahdanauAttention(feature_out, hidden_dim)
        self.drop = nn.Dropout(p=p)
        self.init_h = nn.Linear(feature_out, hidden_dim)

    def _named_tensors(self):
        t = {"feature_fc.0.weight": self.feature_fc[0].weight, "feature_fc.0.bias": self.feature_fc[0].bias,
             "feature_fc.2.weight": self.feature_fc[2].weight, "feature_fc.2.bias": self.feature_fc[2].bias,
             "embed.weight": self.embed.weight, "fc.weight": self.fc.weight, "fc.bias": self.fc.bias,
             "attention.W_a.weight": self.attention.W_a.weight, "attention.W_a.bias": self.attention.W_a.bias,
             "attention.U_a.weight": self.attention.U_a.weight, "attention.U_a.bias": self.attention.U_a.bias,
             "attention.v_a.weight": self.attention.v_a.weight, "attention.v_a.bias": self.attention.v_a.bias,
             "init_h.weight": self.init_h.weight, "init_h.bias": self.init_h.bias}
        # generated (or ordinary) cell weights: Parameters or plain tensors set by set_all_parameters
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            t["gru." + n] = getattr(self.gru, n)
        return t

    def forward(self, features, captions, sample_prob=0.0):
        """features [B,P,num_features], captions [B,T] -> (outputs [B,T,V], atten_weights [B,T,P])."""
        if sample_prob != 0.0:
            raise NotImplementedError("scheduled sampling / free running (sample_prob > 0, "
                                      "models/decoderlstm.py:89-96) is not built yet (SURVEY.md 8f N1)")
        if self.layers:
            raise NotImplementedError("num_layers > 1 is not supported by the fused HIP path")
        if self.training and self.drop.p > 0:
            raise NotImplementedError("dropout p > 0 is not supported by the fused HIP path "
                                      "(the hypernet path constructs AttentionGru with p=0.0)")
        B, P, D = features.shape
        if D != self.num_features:
            raise CaphnError(f"features have {D} channels, module expects {self.num_features}")
        dims = ops.DecDims(B, captions.shape[1], P, D, self.feature_out, self.embedding_dim, self.hidden_dim,
                           self.vocab_size)
        return CF.attention_gru_forward(dims, features, captions, self._named_tensors())

    def init_hidden(self, features):
        """models/decoderlstm.py:122-135 (features are post-feature_fc here, as in the reference)."""
        return self.init_h(torch.mean(features, dim=1))


class GruNet(AttentionGru):
    """Name imported by train_gru.py:12 but defined nowhere in the reference; its constructor call
    (train_gru.py:45) is AttentionGru's and its result is used as a single tensor (:84-86)."""

    def forward(self, features, captions, sample_prob=0.0):
        return super().forward(features, captions, sample_prob)[0]


# names hypernet.py:11 imports from models.decoderlstm (they only exist in the import-less later.py)
DecoderGRU = GruNet
DecoderRNN = GruNet
