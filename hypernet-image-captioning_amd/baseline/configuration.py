"""Transformer hyper-parameters of the reference's baseline/configuration.py:24-36 (values only)."""
from dataclasses import dataclass


@dataclass
class Config:
    hidden_dim: int = 256
    pad_token_id: int = 0
    max_position_embeddings: int = 128
    layer_norm_eps: float = 1e-12
    dropout: float = 0.1
    vocab_size: int = 30522
    enc_layers: int = 6
    dec_layers: int = 6
    dim_feedforward: int = 2048
    nheads: int = 8
    pre_norm: bool = True
