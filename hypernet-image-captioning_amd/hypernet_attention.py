"""HyperNet with the reference's constructor, attributes and forward contract
(hypernet_attention.py:32-121): hn_base + per-parameter heads emit the GRUCell weights of
`captioner` for one style/domain row; forward(x) injects them and returns the captioner.

Beam search (test_step, :242-306) runs batched and device-resident (`beam_search`); the text-metric / wandb
parts of the reference class are out of scope; pytorch_lightning is optional (the class is a plain nn.Module without it).
"""
import torch
from torch import nn
from torch.nn import functional as F

from caphn import config
from caphn import functional as CF
from caphn import ops
from models.decoderlstm import AttentionGru, AttentionLstm
from models.encoder import EncoderCNN
from utils import flip_parameters_to_tensors, set_all_parameters

try:  # the reference subclasses pl.LightningModule; keep that when Lightning is installed
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


def build_hypernet_layers(cell: nn.Module, hyper_emb: int):
    """Head-sizing rule of hypernet_attention.py:55-99 (N = 1, M = 500)."""
    N, M = 1, 500
    hn_base = nn.Sequential(nn.Linear(hyper_emb, N * hyper_emb), nn.LeakyReLU(),
                            nn.Linear(N * hyper_emb, N * hyper_emb), nn.LeakyReLU())
    heads = []
    for name, W in cell.named_parameters():
        if name in ('embed.weight', 'fc_out.weight', 'fc_out.bias'):
            continue
        w_size = len(W.flatten())
        if w_size < N * hyper_emb:
            raise ValueError("head rule branch 'w < hyper_emb' (hypernet_attention.py:78-83) builds "
                             "Linear(he,1)->Linear(w,w), which cannot run; unsupported")
        if w_size // M < N * hyper_emb:
            heads.append(nn.Sequential(nn.Linear(N * hyper_emb, N * hyper_emb), nn.LeakyReLU(),
                                       nn.Linear(N * hyper_emb, w_size)))
        else:
            heads.append(nn.Sequential(nn.Linear(N * hyper_emb, w_size // M), nn.LeakyReLU(),
                                       nn.Linear(w_size // M, w_size)))
    return hn_base, nn.ModuleList(heads)


class HyperNet(_Base):
    def __init__(self, feature_size, embed_size, hidden_size, vocab_size, vocab, num_layers=1, lr=1e-6,
                 mixup=False, alpha=0.3, cc=False, hyper_emb=10, cell='gru'):
        """Reference signature (hypernet_attention.py:33) plus `cell`: 'gru' builds the reference's
        AttentionGru captioner; 'lstm' (extension) builds an AttentionLstm behind the
        same feature_fc and generates the LSTMCell's weights instead."""
        super().__init__()
        hp = self.hparams if pl is not None else {}
        hp['feature_size'] = feature_size; hp['vocab_size'] = vocab_size; hp['embed_size'] = embed_size
        hp['hidden_size'] = hidden_size; hp['lr'] = lr; hp['num_layers'] = num_layers
        if pl is None:
            self.hparams = hp
        self.vocab = vocab
        self.teacher_forcing_proba = 0.0
        self.beam_size = 3
        self.mixup = mixup
        self.alpha = alpha
        self.image_encoder = EncoderCNN()
        self.cell = cell
        if cell == 'gru':
            self.captioner = AttentionGru(2048, feature_size, embed_size, hidden_size, vocab_size, p=0.0)
        elif cell == 'lstm':
            self.captioner = AttentionLstm(2048, embed_size, hidden_size, vocab_size, p=0.0, feature_out=feature_size)
        else:
            raise ValueError("cell must be 'gru' or 'lstm'")
        if not cc:
            hyper_emb = embed_size
        self.hyper_emb = hyper_emb
        self.hn_base, self.hn_heads = build_hypernet_layers(self.cell_module, hyper_emb)
        self._shape = ops.HyperShape(hyper_emb, [(h[0].out_features, h[2].out_features) for h in self.hn_heads])

    @property
    def cell_module(self):
        return self.captioner.gru if self.cell == 'gru' else self.captioner.lstm

    def hyper_named_tensors(self):
        t = {"hn_base.0.weight": self.hn_base[0].weight, "hn_base.0.bias": self.hn_base[0].bias,
             "hn_base.2.weight": self.hn_base[2].weight, "hn_base.2.bias": self.hn_base[2].bias}
        for i, h in enumerate(self.hn_heads):
            t[f"hn_heads.{i}.0.weight"] = h[0].weight; t[f"hn_heads.{i}.0.bias"] = h[0].bias
            t[f"hn_heads.{i}.2.weight"] = h[2].weight; t[f"hn_heads.{i}.2.bias"] = h[2].bias
        return t

    def forward(self, x):
        """hypernet_attention.py:111-121.  x: [1,he] (Flickr) or [he] (CC one-hot row)."""
        heads_out = CF.hyper_forward(self._shape, x, self.hyper_named_tensors(), factor_sink=self.__dict__.get('rank1_factors'))
        flip_parameters_to_tensors(self.cell_module)
        set_all_parameters(self.cell_module, heads_out.reshape(1, -1))
        return self.captioner

    def configure_optimizers(self):
        """hypernet_attention.py:123-133 (parameter order kept)."""
        params = list(self.hn_heads.parameters())
        params.extend(list(self.hn_base.parameters()))
        params.extend(list(self.captioner.feature_fc.parameters()))
        params.extend(list(self.captioner.embed.parameters()))
        params.extend(list(self.captioner.fc.parameters()))
        params.extend(list(self.captioner.attention.parameters()))
        params.extend(list(self.captioner.init_h.parameters()))
        optimizer = config.make_adam(params, self.hparams['lr'], hypernet=self)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, cooldown=2, factor=0.5)
        return [optimizer], [{'scheduler': scheduler, 'monitor': 'val_loss with TF', 'interval': 'epoch'}]

    def configure_gradient_clipping(self, optimizer, optimizer_idx=None, gradient_clip_val=None, gradient_clip_algorithm=None):
        """Lightning's hook (automatic optimisation calls it between backward and optimizer.step; 1.x passes an optimizer index
        as well, 2.x does not): the Trainer's gradient_clip_val (cc_train_hypernet.py:405: 5.0) is applied inside the fused
        optimiser's step, over the dense gradients and the rank-1 members torch's clip_grad_norm_ cannot see."""
        if isinstance(optimizer_idx, float) and gradient_clip_val is None:       # 2.x called positionally
            optimizer_idx, gradient_clip_val, gradient_clip_algorithm = None, optimizer_idx, gradient_clip_val
        if not config.lightning_gradient_clipping(optimizer, gradient_clip_val, gradient_clip_algorithm):
            self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)

    def _style_ids(self, style, dev):
        """torch.tensor([vocab(style)], device=...) of the reference (hypernet_attention.py:139), made once per style and device: a
        tensor built from a Python list is a pageable host-to-device copy the host WAITS for -- 0.4 ms of the 2.5 ms this loop's host
        side takes per step (tools/profile_module_api.py)."""
        cache = self.__dict__.setdefault("_style_id_cache", {})
        key = (style, str(dev))
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.tensor([self.vocab(style)], dtype=torch.long, device=dev)
        return t

    def training_step(self, train_batch, batch_idx):
        """hypernet_attention.py:136-204 without the mixup/BERT and text-metric parts."""
        imgs, (style, (caps, lengths)) = train_batch
        dev = self.captioner.embed.weight.device
        style = self._style_ids(style, dev)
        style_embed = self.captioner.embed(style)
        self.forward(style_embed)
        img_feats = self.image_encoder(imgs.float())
        if self.cell == 'gru':
            caps_pred, _ = self.captioner(img_feats, caps.long(), self.teacher_forcing_proba)
        else:
            caps_pred, _ = self.captioner(caps.long(), img_feats, self.teacher_forcing_proba)
        return CF.cross_entropy(caps_pred.view(-1, self.hparams['vocab_size']), caps.view(-1).long(),
                                ignore_index=self.vocab.w2i['<pad>'])

    def beam_search(self, features, beam_size=None, max_step=50, end_token=None):
        """Beam search of test_step (hypernet_attention.py:251-306) for a batch of encoder outputs [B,P,2048]
        with the currently injected cell weights.  Returns a list with, per image, the best completed sequence
        (token list starting with 0 and ending with </s>) or None when the loop ran past `max_step`
        (the reference's `compute = False`), plus the tensor of their scores."""
        if self.cell != 'gru':
            raise NotImplementedError("beam search follows the reference's GRU captioner")
        cap = self.captioner
        k = self.beam_size if beam_size is None else beam_size
        if end_token is None:
            end_token = self.vocab.w2i['</s>']
        B, P, D = features.shape
        named = cap._named_tensors()
        dims = cap.dec_dims(B, 1, P)
        params = {n: named[n].detach().contiguous().float() for n in dims.names()}
        seqs, lengths, scores, finished, _ = ops.decoder_search(dims, params, features.detach().float().contiguous(), k,
                                                                max_step + 1, end_token=end_token)
        seqs, lengths, fin = seqs.cpu(), lengths.cpu(), finished.cpu()
        return [seqs[b, :int(lengths[b])].tolist() if bool(fin[b]) else None for b in range(B)], scores

    def test_step(self, test_batch, batch_idx):
        """hypernet_attention.py:242-320 without the text metrics: returns the beam-search caption (token list)."""
        imgs, (style, (caps, lengths)) = test_batch
        dev = self.captioner.embed.weight.device
        style = self._style_ids(style, dev)
        self.forward(self.captioner.embed(style))
        with torch.no_grad():
            features = self.image_encoder(imgs.float())
            out, _ = self.beam_search(features)
        return out[0] if len(out) == 1 else out
