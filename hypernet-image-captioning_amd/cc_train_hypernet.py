"""HyperNetCC with the reference's constructor and step contract (cc_train_hypernet.py:42-240): a
domain-embedding front-end (six `embedding` modes, :63-106) feeds ONE style row per minibatch
(`domains[0]`, :136) to the hypernet, which generates the captioner's GRUCell weights.

Out of scope here: text metrics, wandb logging, the data loaders and the
`__main__` driver.  The histogram / tf-idf / JSD features are preprocessing: pass them as
`domain_features` (dict domain -> list of floats), or give `train_file` for the two plain
histogram modes (utils.get_hist_embedding).
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from caphn import config
from caphn import functional as CF

from hypernet_attention import HyperNet, _Base, pl
from models.encoder import EncoderCNN
from utils import get_hist_embedding


class HyperNetCC(_Base):
    def __init__(self, feature_size, embed_size, hidden_size, vocab_size, vocab, list_domain, lr=1e-6, mixup=False,
                 alpha=0.3, hyper_emb=10, embedding='one hot', n_tsne=2, zero_shot=False, list_zeroshot=[],
                 train_file='data/train_cap_100.txt', domain_features=None):
        super().__init__()
        hp = self.hparams if pl is not None else {}
        hp['feature_size'] = feature_size; hp['vocab_size'] = vocab_size; hp['embed_size'] = embed_size
        hp['hidden_size'] = hidden_size; hp['lr'] = lr
        if pl is None:
            self.hparams = hp
        self.vocab = vocab
        self.teacher_forcing_proba = 0.0
        self.beam_size = 3
        self.mixup = mixup
        self.alpha = alpha
        self.embedding = embedding
        self.list_domain = list_domain
        self.epoch = 0
        hist_modes = ('histograme', 'histograme log', 'histograme tfidf')
        if embedding in hist_modes or embedding == 'JSD':
            if domain_features is not None:
                self.dict_domain = {k.replace("\n", ''): list(v) for k, v in domain_features.items()}
            elif embedding in ('histograme', 'histograme log') and os.path.exists(train_file):
                self.dict_domain = get_hist_embedding(train_file, vocab, list_domain, embedding == 'histograme log')
            else:
                raise ValueError(f"embedding={embedding!r} needs precomputed domain_features (tf-idf / JSD-tSNE are "
                                 "preprocessing, reference utils.py:395-441) or an existing train_file")
        else:
            self.dict_domain = {}
            for i in range(len(list_domain)):                               # :83-87
                self.dict_domain[list_domain[i].replace("\n", '')] = i
                for j in range(len(list_zeroshot)):
                    self.dict_domain[list_zeroshot[j].replace("\n", '')] = j + 100
        if embedding == 'one hot':                                           # :89-92
            x = torch.tensor(list(self.dict_domain.values()))
            self.embed = torch.nn.functional.one_hot(x, len(self.dict_domain))
            self.hyper_emb = len(self.dict_domain)
        elif embedding == 'embedding':                                       # :93-95
            self.embed = nn.Embedding(len(self.dict_domain), hyper_emb)
            self.hyper_emb = hyper_emb
        elif embedding in hist_modes:                                        # :96-103
            self.embed = nn.Sequential(nn.Linear(len(vocab) + 1, hyper_emb * 4), nn.LeakyReLU(),
                                       nn.Linear(hyper_emb * 4, hyper_emb), nn.LeakyReLU())
            self.hyper_emb = hyper_emb
        elif embedding == 'JSD':                                             # :104-109
            self.embed = nn.Sequential(nn.Linear(n_tsne, hyper_emb), nn.LeakyReLU())
            self.hyper_emb = hyper_emb
        else:
            raise ValueError(f"unknown embedding mode {embedding!r}")
        self.image_encoder = EncoderCNN()
        self.hypernet = HyperNet(feature_size, embed_size, hidden_size, vocab_size, vocab, num_layers=1, lr=1e-6,
                                 mixup=False, alpha=0.3, cc=True, hyper_emb=self.hyper_emb)

    def _device(self):
        return self.hypernet.captioner.fc.weight.device

    def style_embedding(self, domain):
        """cc_train_hypernet.py:136-149: the hypernet's input row for one domain name."""
        dev = self._device()
        # (the per-domain input tensors are made once per device: built from Python data they are host-to-device copies the host waits for)
        cache = self.__dict__.setdefault("_style_in_cache", {})
        key = (domain, str(dev))
        if self.embedding == 'embedding':
            if key not in cache:
                cache[key] = torch.tensor(self.dict_domain[domain], dtype=torch.long, device=dev)
            return self.embed(cache[key])
        if self.embedding == 'one hot':
            if key not in cache:
                cache[key] = self.embed[self.dict_domain[domain]].to(device=dev, dtype=torch.float32)
            return cache[key]
        if key not in cache:
            cache[key] = torch.tensor(self.dict_domain[domain], dtype=torch.float32, device=dev)
        return self.embed(cache[key])

    def domain_input(self, domain):
        """What FusedTrainer.from_cc(self).step(..., domain_input=...) takes for one domain name: the embedding index, or the
        domain's feature vector on the device ('one hot' has no front-end: pass style_embedding(domain) as x_style)."""
        if self.embedding == 'embedding':
            return int(self.dict_domain[domain])
        if self.embedding == 'one hot':
            raise ValueError("'one hot' has no trainable front-end: use x_style=self.style_embedding(domain)")
        cache = self.__dict__.setdefault('_domain_inputs', {})
        t = cache.get(domain)
        if t is None:
            t = cache[domain] = torch.tensor(self.dict_domain[domain], dtype=torch.float32, device=self._device())
        return t

    def configure_optimizers(self):
        """:110-122 (parameter order kept)."""
        params = list(self.hypernet.hn_heads.parameters())
        if not self.embedding == 'one hot':
            params.extend(list(self.embed.parameters()))
        params.extend(list(self.hypernet.hn_base.parameters()))
        params.extend(list(self.hypernet.captioner.feature_fc.parameters()))
        params.extend(list(self.hypernet.captioner.embed.parameters()))
        params.extend(list(self.hypernet.captioner.fc.parameters()))
        params.extend(list(self.hypernet.captioner.attention.parameters()))
        params.extend(list(self.hypernet.captioner.init_h.parameters()))
        optimizer = config.make_adam(params, self.hparams['lr'], hypernet=self.hypernet)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, cooldown=2, factor=0.5)
        return [optimizer], [{'scheduler': scheduler, 'monitor': 'val_loss with TF', 'interval': 'epoch'}]

    def _loss(self, caps_pred, caps):
        return CF.cross_entropy(caps_pred.view(-1, self.hparams['vocab_size']), caps.view(-1).long(),
                                ignore_index=self.vocab.w2i['<pad>'])

    def configure_gradient_clipping(self, optimizer, optimizer_idx=None, gradient_clip_val=None, gradient_clip_algorithm=None):
        """Lightning's hook (automatic optimisation calls it between backward and optimizer.step; 1.x passes an optimizer index
        as well, 2.x does not): the Trainer's gradient_clip_val (cc_train_hypernet.py:405: 5.0) is applied inside the fused
        optimiser's step, over the dense gradients and the rank-1 members torch's clip_grad_norm_ cannot see."""
        if isinstance(optimizer_idx, float) and gradient_clip_val is None:       # 2.x called positionally
            optimizer_idx, gradient_clip_val, gradient_clip_algorithm = None, optimizer_idx, gradient_clip_val
        if not config.lightning_gradient_clipping(optimizer, gradient_clip_val, gradient_clip_algorithm):
            self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val, gradient_clip_algorithm=gradient_clip_algorithm)

    def training_step(self, train_batch, batch_idx):
        """:134-166 without the text metrics / logging."""
        imgs, caps, lengths, domains = train_batch
        style_embed = self.style_embedding(domains[0])
        captioner = self.hypernet.forward(style_embed)
        img_feats = self.hypernet.image_encoder(imgs.float())
        caps_pred, _ = captioner(img_feats, caps.long(), self.teacher_forcing_proba)
        return self._loss(caps_pred, caps)

    def validation_step(self, val_batch, batch_idx):
        """:169-208: teacher-forced and free-running passes; returns both losses."""
        imgs, caps, lengths, domains = val_batch
        with torch.no_grad():
            style_embed = self.style_embedding(domains[0])
            captioner = self.hypernet.forward(style_embed)
            img_feats = self.hypernet.image_encoder(imgs.float())
            caps_pred_tf, _ = captioner(img_feats, caps.long(), self.teacher_forcing_proba)
            caps_pred, _ = captioner(img_feats, caps.long(), 1.0)
            return {'val_loss': self._loss(caps_pred, caps), 'val_loss with TF': self._loss(caps_pred_tf, caps)}
