"""The CATR transformer of the reference's baseline/transformer.py on libcaphn: same class names, constructor
arguments, call signatures and state_dict keys (a checkpoint of the reference loads with strict=True), every layer
computed by HIP kernels -- projections / feed-forward through caphn_gemm_f32, LayerNorm and the multi-head attention
core through caphn_layernorm_* / caphn_attention_*, residuals through caphn_axpy_f32, the token / position tables
through caphn_embedding_*.  nn.MultiheadAttention / nn.LayerNorm / nn.Linear objects are kept as parameter containers
only (that is what makes the state_dict identical); their forward is never called.

Differences from the reference: in training mode the dropout masks (residual branches, feed-forward, embeddings and the
attention probabilities inside nn.MultiheadAttention) come from libcaphn's counter-based hash (caphn_dropout_f32,
caphn_attn_dims.dropout_p), not from torch's Philox stream: the same regulariser, element-wise different masks, so
parity with the reference is pinned with dropout off and only statistical with it on; `activation` is 'relu' only; one (batch, head) side of an attention must fit LDS
(caphn_attention_supported: 512 positions at 32-wide heads).
Pinned by tests/golden/catr_*.npz, generated from the reference's own classes (tools/make_golden.py --only-catr).
"""
import copy

import torch
from torch import nn

from caphn import functional as CF


def _drop(d: nn.Dropout, x):
    return CF.dropout(x, d.p, d.training)


def _res(d: nn.Dropout, x, branch):
    """x + d(branch) in one kernel."""
    return CF.add(x, branch, d.p, d.training)


def _mha(m: nn.MultiheadAttention, query, key, value, attn_mask=None, key_padding_mask=None):
    """What m(query, key, value, attn_mask=..., key_padding_mask=...)[0] returns   (baseline/transformer.py:137,197-199)."""
    d = m.embed_dim
    w, b = m.in_proj_weight, m.in_proj_bias
    if query is key:                                         # q = k = x + pos: one packed projection, used in place
        qk = CF.linear(query, w[:2 * d], b[:2 * d])
        q, k = qk[..., :d], qk[..., d:]
    else:
        q, k = CF.linear(query, w[:d], b[:d]), CF.linear(key, w[d:2 * d], b[d:2 * d])
    v = CF.linear(value, w[2 * d:], b[2 * d:])
    o = CF.attention(q, k, v, m.num_heads, attn_mask, key_padding_mask, dropout_p=m.dropout if m.training else 0.0)
    return CF.linear(o, m.out_proj.weight, m.out_proj.bias)


def _ln(n: nn.LayerNorm, x):
    return CF.layer_norm(x, n.weight, n.bias, n.eps)


def _with_pos(x, pos):
    return x if pos is None else CF.add(x, pos)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("only activation='relu' (the reference's configuration) is fused")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def _ffn(self, x):
        return CF.ffn(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                      dropout_p=self.dropout.p if self.training else 0.0)

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        if self.normalize_before:                            # baseline/transformer.py:157-168
            x = _ln(self.norm1, src)
            qk = _with_pos(x, pos)
            src = _res(self.dropout1, src, _mha(self.self_attn, qk, qk, x, src_mask, src_key_padding_mask))
            return _res(self.dropout2, src, self._ffn(_ln(self.norm2, src)))
        qk = _with_pos(src, pos)                             # :143-155
        src = _ln(self.norm1, _res(self.dropout1, src, _mha(self.self_attn, qk, qk, src, src_mask, src_key_padding_mask)))
        return _ln(self.norm2, _res(self.dropout2, src, self._ffn(src)))


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("only activation='relu' (the reference's configuration) is fused")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def _ffn(self, x):
        return CF.ffn(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                      dropout_p=self.dropout.p if self.training else 0.0)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        mem_k = _with_pos(memory, pos)
        if self.normalize_before:                            # baseline/transformer.py:229-249
            x = _ln(self.norm1, tgt)
            qk = _with_pos(x, query_pos)
            tgt = _res(self.dropout1, tgt, _mha(self.self_attn, qk, qk, x, tgt_mask, tgt_key_padding_mask))
            x = _ln(self.norm2, tgt)
            tgt = _res(self.dropout2, tgt, _mha(self.multihead_attn, _with_pos(x, query_pos), mem_k, memory, memory_mask,
                                                memory_key_padding_mask))
            return _res(self.dropout3, tgt, self._ffn(_ln(self.norm3, tgt)))
        qk = _with_pos(tgt, query_pos)                       # :205-227
        tgt = _ln(self.norm1, _res(self.dropout1, tgt, _mha(self.self_attn, qk, qk, tgt, tgt_mask, tgt_key_padding_mask)))
        tgt = _ln(self.norm2, _res(self.dropout2, tgt, _mha(self.multihead_attn, _with_pos(tgt, query_pos), mem_k, memory,
                                                           memory_mask, memory_key_padding_mask)))
        return _ln(self.norm3, _res(self.dropout3, tgt, self._ffn(tgt)))


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None, pos=None):
        out = src
        for layer in self.layers:
            out = layer(out, src_mask=mask, src_key_padding_mask=src_key_padding_mask, pos=pos)
        return _ln(self.norm, out) if self.norm is not None else out


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        out, inter = tgt, []
        for layer in self.layers:
            out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask, tgt_key_padding_mask=tgt_key_padding_mask,
                        memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos)
            if self.return_intermediate:
                inter.append(_ln(self.norm, out))
        if self.norm is not None:
            out = _ln(self.norm, out)
            if self.return_intermediate:
                inter[-1] = out
        return torch.stack(inter) if self.return_intermediate else out


class DecoderEmbeddings(nn.Module):
    """baseline/transformer.py:271-300: word + position embedding, LayerNorm(eps = config.layer_norm_eps)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_dim, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_dim)
        self.LayerNorm = nn.LayerNorm(config.hidden_dim, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.dropout)

    def forward(self, x):
        bs, seq = x.shape
        position_ids = torch.arange(seq, dtype=torch.long, device=x.device).unsqueeze(0).expand(bs, seq)
        e = CF.add(CF.embedding(x, self.word_embeddings.weight, self.word_embeddings.padding_idx),
                   CF.embedding(position_ids, self.position_embeddings.weight))
        return _drop(self.dropout, _ln(self.LayerNorm, e))


def generate_square_subsequent_mask(sz, device=None):
    """0 on and below the diagonal, -inf above (baseline/transformer.py:319-326)."""
    return torch.full((sz, sz), float("-inf"), device=device).triu(1)


class Transformer(nn.Module):
    def __init__(self, config, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        enc = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.encoder = TransformerEncoder(enc, num_encoder_layers, nn.LayerNorm(d_model) if normalize_before else None)
        self.embeddings = DecoderEmbeddings(config)
        dec = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.decoder = TransformerDecoder(dec, num_decoder_layers, nn.LayerNorm(d_model), return_intermediate=return_intermediate_dec)
        for p in self.parameters():                          # baseline/transformer.py:35-38
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead = d_model, nhead
        self._causal = {}                                    # (length, device) -> mask, built on the device once

    def forward(self, src, mask, pos_embed, tgt, tgt_mask):
        """src [bs, C, H, W] (projected backbone features), mask [bs, H, W] bool (True = padding), pos_embed like src,
        tgt [bs, T] token ids, tgt_mask [bs, T] bool -> hs [T, bs, C]   (baseline/transformer.py:40-57)."""
        bs = src.shape[0]
        src = src.flatten(2).permute(2, 0, 1).contiguous()                 # HW x bs x C
        pos_embed = pos_embed.flatten(2).permute(2, 0, 1).contiguous()
        mask = mask.flatten(1)
        tgt = self.embeddings(tgt).permute(1, 0, 2).contiguous()
        table = self.embeddings.position_embeddings.weight              # the whole table, repeated over the batch (:50-51)
        ids = torch.arange(table.shape[0], dtype=torch.long, device=table.device).unsqueeze(1).expand(-1, bs)
        query_embed = CF.embedding(ids, table)
        memory = self.encoder(src, src_key_padding_mask=mask, pos=pos_embed)
        key = (len(tgt), tgt.device)
        causal = self._causal.get(key)
        if causal is None:
            causal = self._causal[key] = generate_square_subsequent_mask(len(tgt), tgt.device)
        return self.decoder(tgt, memory, memory_key_padding_mask=mask, tgt_key_padding_mask=tgt_mask, pos=pos_embed,
                            query_pos=query_embed, tgt_mask=causal)


def build_transformer(config):
    return Transformer(config, d_model=config.hidden_dim, dropout=config.dropout, nhead=config.nheads,
                       dim_feedforward=config.dim_feedforward, num_encoder_layers=config.enc_layers,
                       num_decoder_layers=config.dec_layers, normalize_before=config.pre_norm, return_intermediate_dec=False)
