"""HyperNet of the reference's older hypernet.py (:26-114): a wider hn_base (E -> 4E -> 8E) and one head per
parameter of the WHOLE non-attention captioner (DecoderGRU / DecoderRNN of later.py, first cell and extra layers)
except embed / fc_out, sized by the M = 8 rule (:62-93).  forward(x) injects the heads' outputs with
flip_parameters_to_tensors / set_all_parameters on the captioner -- including the reference's child-offset
restart (utils.py:68): every extra layer re-reads theta from offset 0, so its heads' own outputs are never used.

The reference file cannot be imported in the reference itself (hypernet.py:11 names classes that only exist in the
import-less later.py); image encoder (ResNet-101 download), metrics and GloVe loading are out of scope.
"""
import warnings

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from caphn import config
from caphn import functional as CF
from caphn import ops
from models.decoderlstm import DecoderGRU, DecoderRNN
from utils import flip_parameters_to_tensors, set_all_parameters

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


def build_hypernet_layers(captioner: nn.Module, embed_size: int):
    """hypernet.py:55-93."""
    hn_base = nn.Sequential(nn.Linear(embed_size, 4 * embed_size), nn.LeakyReLU(),
                            nn.Linear(4 * embed_size, 8 * embed_size), nn.LeakyReLU())
    heads = []
    for name, W in captioner.named_parameters():
        if name in ('embed.weight', 'fc_out.weight', 'fc_out.bias'):
            continue
        w_size = len(W.flatten())
        if w_size < 8 * embed_size:
            heads.append(nn.Sequential(nn.Linear(8 * embed_size, w_size), nn.LeakyReLU(), nn.Linear(w_size, w_size)))
        elif w_size // 8 < 8 * embed_size:
            heads.append(nn.Sequential(nn.Linear(8 * embed_size, 8 * embed_size), nn.LeakyReLU(),
                                       nn.Linear(8 * embed_size, w_size)))
        else:
            heads.append(nn.Sequential(nn.Linear(8 * embed_size, w_size // 8), nn.LeakyReLU(),
                                       nn.Linear(w_size // 8, w_size)))
    return hn_base, nn.ModuleList(heads)


class HyperNet(_Base):
    def __init__(self, embed_size, hidden_size, vocab_size, vocab, num_layers=1, type='gru', lr=1e-6):
        super().__init__()
        hp = self.hparams if pl is not None else {}
        hp['vocab_size'] = vocab_size; hp['embed_size'] = embed_size; hp['hidden_size'] = hidden_size
        hp['lr'] = lr; hp['num_layers'] = num_layers
        if pl is None:
            self.hparams = hp
        self.vocab = vocab
        self.teacher_forcing_proba = 1.0
        # hypernet.py:38-47: frozen ResNet-101 with a trainable fc -> embed_size.  Out of scope (weights are a download):
        # `image_encoder` maps precomputed 2048-d pooled features (or passes [B, embed_size] rows through)
        self.image_encoder = _PooledEncoder(2048, embed_size)
        if type == 'gru':
            self.captioner = DecoderGRU(embed_size, hidden_size, vocab_size, num_layers=num_layers, dropout=False, vocab=vocab)
        else:
            self.captioner = DecoderRNN(embed_size, hidden_size, vocab_size, num_layers=num_layers, vocab=vocab)
        self.hn_base, self.hn_heads = build_hypernet_layers(self.captioner, embed_size)
        self._shape = ops.HyperShape(8 * embed_size, [(h[0].out_features, h[2].out_features) for h in self.hn_heads],
                                     d_in=embed_size, d_mid=4 * embed_size)

    def hyper_named_tensors(self):
        t = {"hn_base.0.weight": self.hn_base[0].weight, "hn_base.0.bias": self.hn_base[0].bias,
             "hn_base.2.weight": self.hn_base[2].weight, "hn_base.2.bias": self.hn_base[2].bias}
        for i, h in enumerate(self.hn_heads):
            t[f"hn_heads.{i}.0.weight"] = h[0].weight; t[f"hn_heads.{i}.0.bias"] = h[0].bias
            t[f"hn_heads.{i}.2.weight"] = h[2].weight; t[f"hn_heads.{i}.2.bias"] = h[2].bias
        return t

    def forward(self, x):
        """hypernet.py:104-114."""
        heads_out = CF.hyper_forward(self._shape, x, self.hyper_named_tensors(), factor_sink=self.__dict__.get('rank1_factors'))
        flip_parameters_to_tensors(self.captioner)
        set_all_parameters(self.captioner, heads_out.reshape(1, -1))
        return self.captioner

    def configure_optimizers(self):
        """hypernet.py:116-123."""
        params = list(self.hn_heads.parameters())
        params.extend(list(self.hn_base.parameters()))
        params.extend(list(self.captioner.embed.parameters()))
        params.extend(list(self.image_encoder.fc.parameters()))
        optimizer = config.make_adam(params, self.hparams['lr'], hypernet=self)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, cooldown=2)
        return [optimizer], [{'scheduler': scheduler, 'monitor': 'val_loss'}]

    def configure_gradient_clipping(self, optimizer, optimizer_idx=None, gradient_clip_val=None, gradient_clip_algorithm=None):
        """Lightning's hook (automatic optimisation calls it between backward and optimizer.step; 1.x passes an optimizer index
        as well, 2.x does not): the Trainer's gradient_clip_val (cc_train_hypernet.py:405: 5.0) is applied inside the fused
        optimiser's step, over the dense gradients and the rank-1 members torch's clip_grad_norm_ cannot see."""
        if isinstance(optimizer_idx, float) and gradient_clip_val is None:       # 2.x called positionally
            optimizer_idx, gradient_clip_val, gradient_clip_algorithm = None, optimizer_idx, gradient_clip_val
        if not config.lightning_gradient_clipping(optimizer, gradient_clip_val, gradient_clip_algorithm):
            self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)

    def training_step(self, train_batch, batch_idx):
        """hypernet.py:126-152 with teacher forcing (the sampled branch is torch.multinomial per step) and without
        the text metrics.  The loss has no ignore_index here (:146)."""
        imgs, (style, (caps, lengths)) = train_batch
        dev = self.captioner.embed.weight.device
        cache = self.__dict__.setdefault("_style_id_cache", {})       # (a tensor from a Python list is a host-to-device copy the host waits for)
        key = (style, str(dev))
        if key not in cache:
            cache[key] = torch.tensor([self.vocab(style)], dtype=torch.long, device=dev)
        style = cache[key]
        style_embed = self.captioner.embed(style)
        self.forward(style_embed)
        img_feats = self.image_encoder(imgs.float())
        # hypernet.py:134-137: np.random.binomial(1, teacher_forcing_proba) decides between teacher forcing and the sampled
        # branch (a word drawn per step from the previous step's softmax, later.py:424-426)
        teacher_forcing = bool(np.random.binomial(1, self.teacher_forcing_proba))
        caps_pred = self.captioner(img_feats, caps.long(), teacher_forcing)
        loss = CF.cross_entropy(caps_pred.view(-1, self.hparams['vocab_size']), caps.view(-1).long())
        if self.teacher_forcing_proba > 0.25:
            self.teacher_forcing_proba = self.teacher_forcing_proba * 0.9995
        return loss


class _PooledEncoder(nn.Module):
    """Stand-in for hypernet.py:38-47's ResNet-101: only its trainable `fc` (2048 -> embed_size) is kept."""

    def __init__(self, num_ftrs, embed_size):
        super().__init__()
        self.fc = nn.Linear(num_ftrs, embed_size)

    def forward(self, x):
        if x.dim() == 2 and x.shape[1] == self.fc.out_features:
            return x
        if x.dim() == 2 and x.shape[1] == self.fc.in_features:
            return self.fc(x)
        raise ValueError("pass pooled 2048-d ResNet features or ready image embeddings (the ResNet trunk is out of scope)")
