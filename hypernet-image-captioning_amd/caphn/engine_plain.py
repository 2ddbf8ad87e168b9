"""Fused training step of the older hypernet.py (N3): hypernet forward -> DecoderGRU / DecoderRNN forward ->
cross entropy -> backward -> Adam, every arithmetic step a libcaphn kernel.

Call stack it replaces: hypernet.py:126-152 (training_step, teacher forcing) + Lightning's backward + the Adam of
configure_optimizers (:116-123: hn_heads, hn_base, captioner.embed, image_encoder.fc -- fc_out has no optimiser entry;
the Trainer of :218 sets no gradient clipping).  The generated weights stay attached, so the hypernet trains
(caphn.config.DETACH_THETA = False semantics; the literal nn.Parameter re-wrap of utils.py:57 leaves it untrained).

What the byte count is made of at the literal configuration HyperNet(200, 150, 9684, vocab, 2, 'gru') (2.785 G
hypernet parameters, 11.1 GB):
  * set_all_parameters restarts at offset 0 in every child module (utils.py:62-68), so lstm_cell and every extra
    layer read theta[0:n] and the slices produced by the extra layers' own heads are never read.  Those heads get a
    zero gradient, and Adam with zero gradient and zero state is the identity -- so they are left out of the step
    altogether (forward GEMV, VJP and optimiser pass): 1.14 G parameters that the step never touches.
  * the live second-layer weights W2_i [w_i, k_i] are streamed twice: the VJP (4 B/parameter) and one pass that forms
    the rank-1 gradient dtheta_i (x) a_i on the fly, applies Adam to W, m, v (24 B/parameter) and emits the next
    step's theta_i = W2_i' a_i' + b2_i' from the row it has in registers (style known one step ahead).
"""
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import CaphnError


def _up4(n: int) -> int:
    return (n + 3) & ~3


class FusedPlainTrainer:
    def __init__(self, net, lr: float = 1e-6, betas=(0.9, 0.999), eps: float = 1e-8):
        """net: this package's hypernet.HyperNet, already on a CUDA(HIP) device."""
        self.net, self.cap = net, net.captioner
        cap = self.cap
        dev = cap.fc_out.weight.device
        if dev.type != "cuda":
            raise CaphnError("FusedPlainTrainer needs the model on a CUDA(HIP) device")
        if cap.dropout:
            raise NotImplementedError("dropout=True is not fused (hypernet.py:51 passes False)")
        self.dev = dev
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        E, H, self.layers = cap.embed_size, cap.hidden_size, cap.num_layers
        G = (4 if cap._cell == "lstm" else 3) * H
        # theta ranges each child module reads (every child restarts at 0): [(name, numel)] per layer
        self._layer_params = []
        for l in range(self.layers):
            pre = "lstm_cell." if l == 0 else f"layers.{l - 1}."
            self._layer_params.append([(pre + "weight_ih", (G, E if l == 0 else H)), (pre + "weight_hh", (G, H)),
                                       (pre + "bias_ih", (G,)), (pre + "bias_hh", (G,))])
        self._layer_size = [sum(_numel(s) for _, s in lp) for lp in self._layer_params]
        used = max(self._layer_size)
        full: ops.HyperShape = net._shape
        live, o = 0, 0
        for k, w in full.heads:                      # heads whose output range starts inside what is read
            if o < used:
                live += 1
            o += w
        self.n_live = live
        self.shape = ops.HyperShape(full.he, list(full.heads[:live]), d_in=full.d_in, d_mid=full.d_mid)
        self.theta_size = self.shape.theta_size
        if used > self.theta_size:
            raise CaphnError("the hypernet heads produce fewer values than the captioner reads")
        self._build_arena()
        self._acts_layout = ops.hyper_acts_layout(self.shape)
        n_acts = self._acts_layout["_total"][1]
        self._acts = torch.zeros(n_acts, dtype=torch.float32, device=dev)
        self._acts_next = torch.zeros(n_acts, dtype=torch.float32, device=dev)
        self._theta = torch.empty(self.theta_size, dtype=torch.float32, device=dev)
        self._theta_next = torch.empty(self.theta_size, dtype=torch.float32, device=dev)
        self._next_token = None
        self._coef = torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev)       # no clipping in hypernet.py
        self._layer_scratch = [torch.empty(n, dtype=torch.float32, device=dev) for n in self._layer_size[1:]]
        self._frozen_g = {"fc_out.weight": torch.empty_like(cap.fc_out.weight.data),
                          "fc_out.bias": torch.empty_like(cap.fc_out.bias.data)}
        self._hyper_ws = None
        self._bufs: Dict[tuple, dict] = {}
        self._toks: Dict[int, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameter arena
    def _build_arena(self):
        net, cap, dev = self.net, self.cap, self.dev
        hyper = net.hyper_named_tensors()
        order = [(f"hn_heads.{i}.2.bias", hyper[f"hn_heads.{i}.2.bias"]) for i in range(self.n_live)]   # = theta layout
        small = ["hn_base.0.weight", "hn_base.0.bias", "hn_base.2.weight", "hn_base.2.bias"]
        for i in range(self.n_live):
            small += [f"hn_heads.{i}.0.weight", f"hn_heads.{i}.0.bias"]
        order += [(n, hyper[n]) for n in small]
        order += [("captioner.embed.weight", cap.embed.weight), ("image_encoder.fc.weight", net.image_encoder.fc.weight),
                  ("image_encoder.fc.bias", net.image_encoder.fc.bias)]
        offs, o = {}, 0
        for j, (n, p) in enumerate(order):
            if j >= self.n_live:
                o = _up4(o)                          # the theta block itself is contiguous
            offs[n] = (o, p.numel(), tuple(p.shape))
            o += p.numel()
        self.n_dense = _up4(o)
        self.offs = offs
        self.flat_p = torch.zeros(self.n_dense, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        for n, p in order:
            v = self._view(self.flat_p, n)
            v.copy_(p.data)
            p.data = v
            p.grad = self._view(self.flat_g, n)
        self.W2 = [hyper[f"hn_heads.{i}.2.weight"] for i in range(self.n_live)]
        for w in self.W2:
            w.data = w.data.contiguous()
        self.W2_m = [torch.zeros_like(w.data) for w in self.W2]
        self.W2_v = [torch.zeros_like(w.data) for w in self.W2]
        hp = {n: self._view(self.flat_p, n) for n in self.shape.param_names() if n in offs}
        for i in range(self.n_live):
            hp[f"hn_heads.{i}.2.weight"] = self.W2[i].data
        hp["__frozen__"] = True
        self._hp = hp
        self._hg = {n: self._view(self.flat_g, n) for n in small}

    def _view(self, flat, name):
        o, n, shape = self.offs[name]
        return flat[o:o + n].view(shape)

    def _cell_views(self, theta_flat):
        """plain-decoder parameter dict whose cell entries are views of theta (every layer from offset 0)."""
        d = {}
        for lp in self._layer_params:
            o = 0
            for n, shape in lp:
                k = _numel(shape)
                d[n] = theta_flat[o:o + k].view(shape)
                o += k
        return d

    def _buffers(self, B, T):
        b = self._bufs.get((B, T))
        if b is None:
            cap = self.cap
            dims = ops.PlainDims(B, T, cap.embed_size, cap.hidden_size, cap.vocab_size, cap.num_layers, cap._cell)
            b = {"dims": dims, "ws": ops.plain_workspace(dims, self.dev),
                 "logits": torch.empty(B, T, dims.V, dtype=torch.float32, device=self.dev),
                 "dfeat": torch.empty(B, dims.E, dtype=torch.float32, device=self.dev),
                 "feats": torch.empty(B, dims.E, dtype=torch.float32, device=self.dev)}
            self._bufs[(B, T)] = b
        return b

    # ------------------------------------------------------------------ one step
    def step(self, imgs, captions, style_token: int, h0: Optional[torch.Tensor] = None, c0: Optional[torch.Tensor] = None,
             next_style_token: Optional[int] = None, teacher_forcing: bool = True, seed: Optional[int] = None):
        """imgs: [B, 2048] pooled ResNet features (through image_encoder.fc) or ready [B, E] embeddings; captions
        [B, T] int64; style_token: vocab id of the style word (x = captioner.embed.weight[token], hypernet.py:128-131).
        h0 / c0: initial states; default as the module draws them (DecoderGRU: torch.rand on the CPU generator).
        next_style_token: the next minibatch's style when the loader is one batch ahead -- the optimiser pass then
        emits that step's theta.  teacher_forcing=False: hypernet.py:135-140's other branch (later.py:418-431) -- from step 1 on
        the cell is fed a word drawn from the previous step's softmax (caphn_plain_forward_sampled: counter-based draws from
        `seed`, default a fresh one per step; the ids are kept in self.last_chosen [B, T], -1 at step 0); the loss is still
        taken against the caption and nothing flows through the draw.  Returns the device tensor [loss, n_targets] (reused by
        the next call)."""
        cap, dev = self.cap, self.dev
        B, T = captions.shape
        buf = self._buffers(B, T)
        dims = buf["dims"]
        self.step_count += 1
        ops.zero_(self.flat_g)
        if h0 is None:
            h0, c0 = cap._initial_state(B, imgs)
        embed_w = self._view(self.flat_p, "captioner.embed.weight")
        # ---- hypernet forward (or the theta the previous optimiser pass already produced)
        if self._next_token is not None and self._next_token == int(style_token):
            self._theta, self._theta_next = self._theta_next, self._theta
            self._acts, self._acts_next = self._acts_next, self._acts
        else:
            ops.hyper_forward(self.shape, self._hp, embed_w[int(style_token)], theta=self._theta, acts=self._acts)
        self._next_token = None
        theta = self._theta
        # ---- decoder forward
        params = self._cell_views(theta)
        params.update({"fc_out.weight": cap.fc_out.weight.data, "fc_out.bias": cap.fc_out.bias.data, "embed.weight": embed_w})
        through_fc = imgs.shape[1] != dims.E
        if through_fc:
            feats = ops.gemm(imgs, self._view(self.flat_p, "image_encoder.fc.weight"), tb=True,
                             bias=self._view(self.flat_p, "image_encoder.fc.bias"), out=buf["feats"])
        else:
            feats = imgs
        if teacher_forcing:
            logits = ops.plain_forward(dims, params, feats, captions, h0, c0, buf["ws"], check_ids=False, logits=buf["logits"])
            self.last_chosen = None
        else:
            if seed is None:
                from .functional import next_seed
                seed = next_seed()
            # (the workspace keeps the drawn ids: caphn_plain_backward scatters d x_t into THEIR embedding rows)
            logits, self.last_chosen = ops.plain_forward_sampled(dims, params, feats, h0, c0, buf["ws"], seed)
        loss, dlogits = ops.cross_entropy_fwd_bwd(logits, captions, ignore_index=-100, dlogits=logits)
        # ---- decoder backward: the first cell's gradients land in dtheta directly, the extra layers' (views of the
        # same range) in scratch and are then added
        dtheta = self.flat_g[:self.theta_size]
        grads = {}
        o = 0
        for n, shape in self._layer_params[0]:
            k = _numel(shape)
            grads[n] = dtheta[o:o + k].view(shape)
            o += k
        for l in range(1, self.layers):
            o = 0
            for n, shape in self._layer_params[l]:
                k = _numel(shape)
                grads[n] = self._layer_scratch[l - 1][o:o + k].view(shape)
                o += k
        grads.update(self._frozen_g)
        grads["embed.weight"] = self._view(self.flat_g, "captioner.embed.weight")
        ops.plain_backward(dims, params, feats, captions, h0, c0, dlogits, grads, buf["ws"], buf["dfeat"])
        for l in range(1, self.layers):
            n = self._layer_size[l]
            ops.axpy_(dtheta[:n], self._layer_scratch[l - 1])
        if through_fc:
            ops.gemm(buf["dfeat"], imgs, ta=True, out=self._view(self.flat_g, "image_encoder.fc.weight"))
            ops.colsum(buf["dfeat"], out=self._view(self.flat_g, "image_encoder.fc.bias"))
        # ---- hypernet VJP (second-layer weight gradients stay rank-1: dtheta_i (x) a_i)
        if self._hyper_ws is None:
            import ctypes as C
            from . import _lib as L
            d = ops._hyper_desc(self.shape, self._hp)
            self._hyper_ws = torch.empty(L.load().caphn_hyper_backward_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        gx = ops.hyper_backward(self.shape, self._hp, dtheta, self._acts, self._hg, want_x=True, ws=self._hyper_ws)
        tok = self._toks.get(int(style_token))
        if tok is None:
            tok = self._toks[int(style_token)] = torch.full((1,), int(style_token), dtype=torch.int64, device=dev)
        ops.embedding_scatter_add(gx.view(1, -1), tok, grads["embed.weight"])
        # ---- Adam
        step = self.step_count
        ops.adam_dense(self.flat_p, self.flat_m, self.flat_v, self.flat_g, self._coef, self.lr, step, self.betas, self.eps)
        prefetch = next_style_token is not None
        if prefetch:
            ops.hyper_forward_acts(self.shape, self._hp, embed_w[int(next_style_token)], self._acts_next)
            self._next_token = int(next_style_token)
        o = 0
        for i, (k, w) in enumerate(self.shape.heads):
            ao, an = self._acts_layout[f"a{i}"]
            kw = {}
            if prefetch:
                kw = dict(next_a=self._acts_next[ao:ao + an], next_bias=self._view(self.flat_p, f"hn_heads.{i}.2.bias"),
                          next_theta=self._theta_next[o:o + w])
            ops.adam_rank(self.W2[i].data, self.W2_m[i], self.W2_v[i], dtheta[o:o + w].view(1, -1),
                          self._acts[ao:ao + an].view(1, -1), self._coef, self.lr, step, self.betas, self.eps, **kw)
            o += w
        return loss


    # ------------------------------------------------------------------ checkpoint / resume
    def optimizer_param_names(self) -> List[str]:
        """hypernet.py:116-123: hn_heads (ALL heads, also the ones whose output nothing reads), hn_base, captioner.embed,
        image_encoder.fc."""
        names = []
        for i in range(len(self.net._shape.heads)):
            names += [f"hn_heads.{i}.0.weight", f"hn_heads.{i}.0.bias", f"hn_heads.{i}.2.weight", f"hn_heads.{i}.2.bias"]
        return names + ["hn_base.0.weight", "hn_base.0.bias", "hn_base.2.weight", "hn_base.2.bias", "captioner.embed.weight",
                        "image_encoder.fc.weight", "image_encoder.fc.bias"]

    def _moments(self, name):
        if name.startswith("hn_heads."):
            i = int(name.split(".")[1])
            if i >= self.n_live:
                return None                      # never receives a gradient: torch's Adam keeps no state for it either
            if name.endswith(".2.weight"):
                return self.W2_m[i], self.W2_v[i]
        return self._view(self.flat_m, name), self._view(self.flat_v, name)

    def state_dict(self) -> dict:
        """torch.optim.Adam's state_dict layout over the reference's parameter list (see FusedTrainer.state_dict)."""
        names = self.optimizer_param_names()
        state = {}
        for i, n in enumerate(names):
            mv = self._moments(n)
            if mv is not None and self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": mv[0].detach().clone(),
                            "exp_avg_sq": mv[1].detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "param_names": names}

    def load_state_dict(self, sd: dict) -> None:
        names = self.optimizer_param_names()
        if "param_names" in sd and list(sd["param_names"]) != names:
            raise CaphnError("optimiser state was saved for a different parameter list")
        steps = set()
        for i, n in enumerate(names):
            mv = self._moments(n)
            if mv is None or i not in sd["state"]:
                continue
            e = sd["state"][i]
            mv[0].copy_(e["exp_avg"].to(device=self.dev, dtype=torch.float32).view_as(mv[0]))
            mv[1].copy_(e["exp_avg_sq"].to(device=self.dev, dtype=torch.float32).view_as(mv[1]))
            steps.add(int(e["step"]))
        if len(steps) > 1:
            raise CaphnError(f"per-parameter step counts differ ({sorted(steps)})")
        self.step_count = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self._next_token = None                  # a theta prefetched from the previous weights is stale


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n
