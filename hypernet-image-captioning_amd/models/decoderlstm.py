"""Decoder models with the reference's class names, constructor signatures, sub-module names
and return types (models/decoderlstm.py:11-185).  Sub-modules stay ordinary nn.Modules holding
ordinary nn.Parameters in the reference's layout, so callers can call, swap and checkpoint them
(hypernet_attention.py:251-272, :419-428); the teacher-forced forward reads their storage and
runs entirely in libcaphn's HIP kernels."""
import numpy as np
import torch
from torch import nn

from caphn import functional as CF
from caphn import ops
from caphn._lib import CaphnError
from .attention import BahdanauAttention


def _sampled_forward(module, dims, features, captions, sample_prob):
    """Scheduled sampling / free running (models/decoderlstm.py:78-96, :236-251).  The per-timestep
    draws consume numpy's global RNG exactly like the reference: one np.random.random() per step, compared
    with 0.0 at t = 0 and with sample_prob afterwards.  Under torch.no_grad() (validation_step,
    cc_train_hypernet.py:187-188) nothing is kept; with gradients enabled (train_gru.py:84 trains through
    sample_prob = 1.0) the forward keeps the backward state and the node is differentiable."""
    flags = [bool(np.random.random() < (0.0 if t == 0 else sample_prob)) for t in range(dims.T)]
    if not features.is_cuda:
        raise CaphnError("libcaphn's HIP kernels need CUDA(HIP) tensors (there is no CPU fallback)")
    named = module._named_tensors()
    if torch.is_grad_enabled() and any(t.requires_grad for t in named.values()):
        return CF.attention_sampled_forward(dims, flags, features, captions, named)
    params = {n: named[n].detach().contiguous().float() for n in dims.names()}
    ws = ops.decoder_workspace(dims, features.device)
    return ops.decoder_forward_sampled(dims, params, features.detach().float().contiguous(),
                                       captions.long().contiguous(), flags, ws)


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
            for l in range(self.num_layers - 1):       # the extra cells h = layer(h, h) (models/decoderlstm.py:34-36)
                t[f"layers.{l}.{n}"] = getattr(self.layers[l], n)
        return t

    def forward(self, features, captions, sample_prob=0.0):
        """features [B,P,num_features], captions [B,T] -> (outputs [B,T,V], atten_weights [B,T,P])."""
        B, P, D = features.shape
        if D != self.num_features:
            raise CaphnError(f"features have {D} channels, module expects {self.num_features}")
        dims = self.dec_dims(B, captions.shape[1], P)
        if sample_prob != 0.0:
            if self.training and self.drop.p > 0:
                raise NotImplementedError("free running / scheduled sampling with dropout active is not fused: call .eval()")
            return _sampled_forward(self, dims, features, captions, sample_prob)
        if self.training and self.drop.p > 0:
            # h = self.drop(h) (models/decoderlstm.py:104): counter-based mask, a fresh seed per call (the backward re-derives it)
            import dataclasses
            dims = dataclasses.replace(dims, drop_p=float(self.drop.p), seed=CF.next_seed())
        return CF.attention_gru_forward(dims, features, captions, self._named_tensors())

    def dec_dims(self, B, T, P):
        return ops.DecDims(B, T, P, self.num_features, self.feature_out, self.embedding_dim, self.hidden_dim,
                           self.vocab_size, layers=max(1, self.num_layers))

    def init_hidden(self, features):
        """models/decoderlstm.py:122-135 (features are post-feature_fc here, as in the reference)."""
        return self.init_h(torch.mean(features, dim=1))

    def greedy_search(self, features, end_sentence=2, max_sentence=20):
        """models/decoderlstm.py:138-175.  features are feature_fc outputs [1,P,F] -> (sentence, weights): the
        token list (ending with end_sentence unless max_sentence was hit) and one attention map [1,P] per step.
        A batch [B,P,F] decodes every image at once on the device and returns lists of those per image."""
        B, P, Fo = features.shape
        if Fo != self.feature_out:
            raise CaphnError(f"greedy_search takes feature_fc outputs ({self.feature_out} channels), got {Fo}")
        named = self._named_tensors()
        dims = ops.DecDims(B, 1, P, Fo, Fo, self.embedding_dim, self.hidden_dim, self.vocab_size, raw=True,
                           layers=max(1, self.num_layers))
        params = {n: named[n].detach().contiguous().float() for n in dims.names()}
        seqs, lengths, _, _, alphas = ops.decoder_search(dims, params, features.detach().float().contiguous(), 1,
                                                         max_sentence, end_token=end_sentence, greedy=True,
                                                         want_alphas=True)
        seqs, lengths = seqs.cpu(), lengths.cpu()
        sents = [seqs[b, 1:int(lengths[b])].tolist() for b in range(B)]
        wts = [[alphas[b:b + 1, t] for t in range(len(sents[b]))] for b in range(B)]
        return (sents[0], wts[0]) if B == 1 else (sents, wts)

    def infer(self, features, end_sentence=2, max_len=40, vocab=None):
        """models/decoderlstm.py:178-185.  The reference unpickles data/vocab.pkl on every call; pass the
        vocabulary object (anything with an `i2w` dict) instead."""
        from utils import clean_sentence
        if vocab is None:
            raise ValueError("pass vocab= (the reference reads data/vocab.pkl here; this port does not unpickle files)")
        feats = self.feature_fc(features)
        output, _ = self.greedy_search(feats, end_sentence, max_len)
        return clean_sentence(output, vocab)


class GruNet(AttentionGru):
    """Name imported by train_gru.py:12 but defined nowhere in the reference; its constructor call
    (train_gru.py:45) is AttentionGru's and its result is used as a single tensor (:84-86)."""

    def forward(self, features, captions, sample_prob=0.0):
        return super().forward(features, captions, sample_prob)[0]


class AttentionLstm(nn.Module):
    """Reference AttentionLstm (models/decoderlstm.py:188-261): constructor
    (num_features, embedding_dim, hidden_dim, vocab_size, p=0.5), sub-modules `embeddings`, `lstm`,
    `fc`, `attention`, `drop`, `init_h`, `init_c`, forward(captions, features, sample_prob=1.0).
    Attention runs directly over the num_features-channel features (no feature_fc).

    feature_out (extension): when given, a feature_fc
    Linear(num_features, feature_out)+ReLU+Linear(feature_out, feature_out) precedes the attention, as
    in AttentionGru -- the configuration the hypernet-LSTM path uses (its G slab then fits the LDS)."""

    def __init__(self, num_features, embedding_dim, hidden_dim, vocab_size, p=0.5, feature_out=None):
        super().__init__()
        self.num_features = num_features
        self.embedding_dim = embedding_dim
        self.hidden_dim = hidden_dim
        self.vocab_size = vocab_size
        self.sample_temp = 0.5
        self.feature_out = feature_out
        att_in = num_features
        if feature_out is not None:
            self.feature_fc = nn.Sequential(nn.Linear(num_features, feature_out), nn.ReLU(),
                                            nn.Linear(feature_out, feature_out))
            att_in = feature_out
        self.embeddings = nn.Embedding(vocab_size, embedding_dim)
        self.lstm = nn.LSTMCell(embedding_dim + att_in, hidden_dim)
        self.fc = nn.Linear(hidden_dim, vocab_size)
        self.attention = BahdanauAttention(att_in, hidden_dim)
        self.drop = nn.Dropout(p=p)
        self.init_h = nn.Linear(att_in, hidden_dim)
        self.init_c = nn.Linear(att_in, hidden_dim)

    @property
    def embed(self):            # uniform access for the fused engine
        return self.embeddings

    def _named_tensors(self):
        t = {"embed.weight": self.embeddings.weight, "fc.weight": self.fc.weight, "fc.bias": self.fc.bias,
             "attention.W_a.weight": self.attention.W_a.weight, "attention.W_a.bias": self.attention.W_a.bias,
             "attention.U_a.weight": self.attention.U_a.weight, "attention.U_a.bias": self.attention.U_a.bias,
             "attention.v_a.weight": self.attention.v_a.weight, "attention.v_a.bias": self.attention.v_a.bias,
             "init_h.weight": self.init_h.weight, "init_h.bias": self.init_h.bias,
             "init_c.weight": self.init_c.weight, "init_c.bias": self.init_c.bias}
        if self.feature_out is not None:
            t.update({"feature_fc.0.weight": self.feature_fc[0].weight, "feature_fc.0.bias": self.feature_fc[0].bias,
                      "feature_fc.2.weight": self.feature_fc[2].weight, "feature_fc.2.bias": self.feature_fc[2].bias})
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            t["lstm." + n] = getattr(self.lstm, n)
        return t

    def dec_dims(self, B, T, P):
        F = self.feature_out if self.feature_out is not None else self.num_features
        return ops.DecDims(B, T, P, self.num_features, F, self.embedding_dim, self.hidden_dim, self.vocab_size,
                           cell="lstm", raw=self.feature_out is None)

    def forward(self, captions, features, sample_prob=1.0):
        """captions [B,T], features [B,P,num_features] -> (outputs [B,T,V], atten_weights [B,T,P]).
        NOTE the reference's argument order and its default sample_prob=1.0 (decoderlstm.py:224)."""
        B, P, D = features.shape
        if D != self.num_features:
            raise CaphnError(f"features have {D} channels, module expects {self.num_features}")
        dims = self.dec_dims(B, captions.shape[1], P)
        if sample_prob != 0.0:
            if self.training and self.drop.p > 0:
                raise NotImplementedError("free running / scheduled sampling with dropout active is not fused: call .eval()")
            return _sampled_forward(self, dims, features, captions, sample_prob)
        if self.training and self.drop.p > 0:       # h = self.drop(h) (models/decoderlstm.py:254), the reference default p = 0.5
            import dataclasses
            dims = dataclasses.replace(dims, drop_p=float(self.drop.p), seed=CF.next_seed())
        return CF.attention_gru_forward(dims, features, captions, self._named_tensors())

    def init_hidden(self, features):
        mean_annotations = torch.mean(features, dim=1)
        return self.init_h(mean_annotations), self.init_c(mean_annotations)


# names hypernet.py:11 imports from models.decoderlstm (they only exist in the import-less later.py)
class _PlainDecoder(nn.Module):
    """Shared body of DecoderGRU / DecoderRNN (later.py:227-457; hypernet.py:11 imports them from this module):
    sub-modules `lstm_cell`, `layers`, `fc_out`, `embed`, `softmax`; forward(features[B,E], captions[B,T],
    teacher_forcing=True) -> outputs [B,T,V]."""
    _cell = "gru"

    def __init__(self, embed_size, hidden_size, vocab_size, num_layers=1, dropout=False, vocab=None):
        super().__init__()
        self.embed_size = embed_size
        self.hidden_size = hidden_size
        self.vocab_size = vocab_size
        self.dropout = dropout
        self.num_layers = num_layers
        self.vocab = vocab            # the reference unpickles data/vocab.pkl here (later.py:372-373); only used for text dumps
        Cell = nn.GRUCell if self._cell == "gru" else nn.LSTMCell
        self.lstm_cell = Cell(input_size=embed_size, hidden_size=hidden_size)          # named lstm_cell in both classes
        self.layers = None
        if num_layers > 1:
            self.layers = nn.ModuleList([Cell(input_size=hidden_size, hidden_size=hidden_size) for _ in range(num_layers - 1)])
        self.fc_out = nn.Linear(in_features=hidden_size, out_features=vocab_size)
        self.embed = nn.Embedding(num_embeddings=vocab_size, embedding_dim=embed_size)
        self.softmax = nn.Softmax(dim=1)

    def _named_tensors(self):
        t = {"fc_out.weight": self.fc_out.weight, "fc_out.bias": self.fc_out.bias, "embed.weight": self.embed.weight}
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            t["lstm_cell." + n] = getattr(self.lstm_cell, n)
            for l in range(self.num_layers - 1):
                t[f"layers.{l}.{n}"] = getattr(self.layers[l], n)
        return t

    def _initial_state(self, batch_size, like):
        raise NotImplementedError

    def forward(self, features, captions, teacher_forcing=True):
        """teacher_forcing=False (later.py:418-431 / :290-301): from step 1 on the cell's input is the embedding of a word drawn
        from softmax(out_{t-1}) -- torch.multinomial in the reference, the kernel's counter-based inverse-CDF draw here (same
        distribution, another random stream); the ids fed are kept in self.last_sampled [B,T]."""
        if self.dropout:
            raise NotImplementedError("dropout=True is not supported by the fused HIP path (hypernet.py:51 passes False)")
        B, T = captions.shape
        dims = ops.PlainDims(B, T, self.embed_size, self.hidden_size, self.vocab_size, self.num_layers, self._cell)
        h0, c0 = self._initial_state(B, features)
        if teacher_forcing:
            return CF.plain_decoder_forward(dims, features, captions, h0, c0, self._named_tensors())
        sample = {"seed": CF.next_seed()}
        out = CF.plain_decoder_forward(dims, features, captions, h0, c0, self._named_tensors(), sample=sample)
        self.last_sampled = sample["chosen"]
        return out


class DecoderGRU(_PlainDecoder):
    """later.py:362-457.  The initial hidden state is torch.rand on the CPU generator, as in the reference (:397-398)."""
    _cell = "gru"

    def _initial_state(self, batch_size, like):
        h0 = torch.rand(size=(batch_size, self.hidden_size)).to(device=like.device, dtype=torch.float32)
        return h0, None


class DecoderRNN(_PlainDecoder):
    """later.py:227-330 (LSTM cells, zero initial states :259-262)."""
    _cell = "lstm"

    def _initial_state(self, batch_size, like):
        z = torch.zeros(batch_size, self.hidden_size, device=like.device, dtype=torch.float32)
        return z, z.clone()
