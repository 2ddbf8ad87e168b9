"""Fused training step: hypernet forward -> decoder forward -> cross entropy -> backward ->
gradient exchange -> global-norm clip -> Adam, every arithmetic step a libcaphn kernel.

Call stack it replaces: HyperNetCC.training_step + Lightning's backward / clip_grad_norm_(5.0) /
Adam.step (cc_train_hypernet.py:134-166, :120, :405; hypernet_attention.py:136-204).

Memory layout (sized for HBM streaming):
  * "dense" arena: one flat fp32 buffer each for parameters, gradients, Adam m and v.  It starts
    with the four second-layer biases in head order -- exactly the layout of theta -- so the
    decoder backward writes dL/dtheta straight into that region: it is at once the bias gradient
    and the row factor of the rank-1 weight gradients.  Then hn_base, the heads' first layers and
    the captioner's non-generated parameters.  Module parameters are re-pointed at arena views,
    so state_dict / sub-module calls keep working and nothing is ever copied per step.
  * the big second-layer weights W2_i [w_i, k_i] keep their own tensors plus m, v; their gradient
    dtheta_i (x) a_i is never materialised: caphn_adam_rank_f32 forms it on the fly while streaming
    W, m, v once (24 B/parameter).
"""
from typing import Dict, List, Optional, Tuple

import os

import torch

from . import dp, ops
from ._lib import CaphnError

def _up4(n: int) -> int:
    return (n + 3) & ~3


class FusedTrainer:
    def __init__(self, hypernet, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: float = 5.0, group=None, frontend=None, frontend_mode: Optional[str] = None):
        """hypernet: a hypernet_attention.HyperNet (this package's) already on a CUDA device.
        frontend / frontend_mode: HyperNetCC's trainable domain front-end `self.embed` and its `embedding` mode
        (cc_train_hypernet.py:93-109: 'embedding', 'histograme', 'histograme log', 'histograme tfidf', 'JSD'; 'one hot' has
        no parameters and needs none of this): its parameters join the arena and the Adam pass, the hypernet's input row is
        computed from `domain_input` by libcaphn GEMMs with a LeakyReLU epilogue, and the row's gradient is pushed back
        through it (from_cc() builds the trainer from a HyperNetCC)."""
        self.net = hypernet
        self.frontend = frontend
        self.fe_mode = frontend_mode
        if frontend is not None and frontend_mode not in ("embedding", "histograme", "histograme log", "histograme tfidf", "JSD"):
            raise CaphnError(f"front-end mode {frontend_mode!r} has no trainable parameters or is unknown")
        self._fe_slot, self._fe_slot_next = {}, {}
        self.cap = hypernet.captioner
        self.shape: ops.HyperShape = hypernet._shape
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.group = group
        self.step_count = 0
        # the loss ignores <pad> targets (ignore_index = 0): their logits / d logits rows are dead work, so the three
        # vocab GEMMs only touch live rows.  Loss and every gradient are unchanged; logits rows of ignored targets
        # are simply never produced (the engine does not return logits).
        self.skip_ignored_rows = True
        dev = self.cap.fc.weight.device
        if dev.type != "cuda":
            raise CaphnError("FusedTrainer needs the model on a CUDA(HIP) device")
        self.dev = dev
        # AttentionGru(num_layers > 1) (models/decoderlstm.py:34-36, :101-103; train_gru.py's default is 2): the extra cells'
        # parameters are ordinary trainable tensors -- they join the arena, the clip norm and the dense Adam pass; the composites
        # run the time loop in per-step windows for them (csrc/decoder_layers.hip)
        d0 = self.cap.dec_dims(1, 1, 1)
        self._cell_names = d0.cell_names()
        # arena order of the decoder's parameters = the order their gradients are finished by the backward, so that
        # the data-parallel all-reduce can go bucket by bucket: vocabulary projection, embedding table, the rest
        # ... and what the backward finishes LAST (feature_fc, attention, init_h: the end of the attention / feature_fc chain) sits
        # right behind the hypernet's small layers, whose VJP ends at about the same time: ONE all-reduce covers both
        back = ["fc.weight", "fc.bias", "embed.weight"]
        self._dec_names = [n for n in d0.names() if n not in self._cell_names and n not in back] + back
        self._vcache = {}
        self._versions = {}
        self._invalidate_caches()
        self._build_arena()
        self._bufs: Dict[Tuple[int, int, int], dict] = {}
        nh = len(self.shape.heads)
        self._acc = torch.zeros(1, dtype=torch.float64, device=dev)
        self._coef = torch.zeros(2, dtype=torch.float32, device=dev)
        self._part = torch.empty(L_sumsq_blocks(self.n_dense), dtype=torch.float64, device=dev)
        self._gradnorm = {}
        self._acts_layout = ops.hyper_acts_layout(self.shape)
        self._acts = torch.zeros(self._acts_layout["_total"][1], dtype=torch.float32, device=dev)
        self._hyper_ws = None
        self._readopted = False
        self._toks = {}      # style token id -> device index tensor
        self._nh = nh
        # step-dependent Adam scalars live in device memory so a captured hipGraph replays for any step
        self._adam_dev = torch.zeros(2, dtype=torch.float32, device=dev)
        # ring of pinned slots: the asynchronous H2D copy reads its slot when the GPU gets there, and the host may be
        # several steps ahead by then -- one slot per step (mod 64) instead of one buffer rewritten every step
        self._adam_host = torch.zeros(64, 2, dtype=torch.float32).pin_memory()
        self._adam_copied = [None] * 64  # event after each slot's H2D copy: a slot is rewritten only once its copy has executed
        self._graph_scalars = False      # eager steps pass the Adam scalars by value (no copy, nothing to race with)
        self._graphs: Dict[tuple, object] = {}
        self._seen = set()
        # next-step theta produced during the optimiser pass (fused GEMV in caphn_adam_rank_gemv_f32)
        self._theta_next = None
        self._acts_next = torch.zeros_like(self._acts)
        self._next_key = None
        # next-step feature_fc / init_hidden / W_a f issued on a side stream beside the small Adam passes
        self._pre_stream = torch.cuda.Stream(device=dev)
        self._fork_ev = None
        self._pre_token = None
        self.fork_late = os.environ.get("CAPHN_FORK_LATE", "1") == "1"
        self._pre_key = None
        self._pre_hold = None
        self._next_hold = None
        self._theta_pre = None
        self._loss_done = torch.cuda.Event()
        self._loss_pending = False
        self._pre_done = torch.cuda.Event()
        # data-parallel exchange: collectives are issued from this stream, which waits on the backward's milestones
        self._comm_stream = torch.cuda.Stream(device=dev)
        self._works = []
        self._xstat = {"steps": 0, "collectives": 0, "timed": []}      # collective_report()
        self._gfac_all = None
        self._acts_all = None
        self.overlap_after_head = 0      # the side stream forks after this head's Adam pass (-1: before the first)
        self.overlap_level = int(os.environ.get("CAPHN_OVERLAP_LEVEL", "4"))
        # 1: the theta-independent front (feature_fc / init_hidden / W_a f) forked after the head-0 pass.
        # 2: the WHOLE front, forked after the W_ih / b_ih / b_hh passes, beside the W_hh pass (round 2: no gain -- feature_fc does
        #    not fit beside one 150 us pass).
        # 3 (default): passes in the order W_ih, b_ih, b_hh, W_hh; the theta-independent front AND G fork after the W_ih pass, the
        #    x-side gate GEMM goes to the caption-side stream behind the b_ih pass: the next forward starts at the recurrent
        #    kernel's prep launch (the two GEMMs that need theta no longer sit between the last pass and the recurrent kernel)
        self._bih_ev = None
        self._aux_pending = False
        self._main_ev = self._zero_lo_done = None
        self.zero_by_adam = os.environ.get("CAPHN_ZERO_BY_ADAM", "1") == "1"
        self._zero_lo_by_adam = False
        self._aux_joined = False
        self._acts_ev = None
        self.pad_logits = os.environ.get("CAPHN_PAD_LOGITS", "1") == "1"
        self.fork_after_acts = os.environ.get("CAPHN_FORK_AFTER_ACTS", "1") == "1"
        self.bias_heads_aside = os.environ.get("CAPHN_BIAS_HEADS_ASIDE", "1") == "1"
        self._pair_packed = None
        self.pack_in_adam = os.environ.get("CAPHN_PACK_IN_ADAM", "1") == "1"
        self._fork0_ev = None
        self.ctx_in_forward = os.environ.get("CAPHN_CTX_IN_FORWARD", "1") == "1"

    # ------------------------------------------------------------------ parameter arenas
    def _build_arena(self):
        net, cap = self.net, self.cap
        hyper = net.hyper_named_tensors()
        dec = {n: t for n, t in cap._named_tensors().items() if n in self._dec_names}
        nh = len(self.shape.heads)
        order: List[Tuple[str, torch.nn.Parameter]] = []
        for i in range(nh):                                   # theta-ordered second-layer biases
            order.append((f"hn_heads.{i}.2.bias", hyper[f"hn_heads.{i}.2.bias"]))
        self.theta_size = sum(p.numel() for _, p in order)
        small = ["hn_base.0.weight", "hn_base.0.bias", "hn_base.2.weight", "hn_base.2.bias"]
        for i in range(nh):
            small += [f"hn_heads.{i}.0.weight", f"hn_heads.{i}.0.bias"]
        for n in small:
            order.append((n, hyper[n]))
        self._hyper_small_end = None
        offs, o = {}, 0
        for k, (n, p) in enumerate(order):
            # the theta block must be contiguous (no padding between the four biases)
            if k >= nh:
                o = _up4(o)
            offs[n] = (o, p.numel(), tuple(p.shape))
            o += p.numel()
        self._hyper_small_end = _up4(o)
        o = self._hyper_small_end
        for n in self._dec_names:
            p = dec[n]
            offs["captioner." + n] = (o, p.numel(), tuple(p.shape))
            o = _up4(o + p.numel())
        fe = self._frontend_tensors()
        for n, prm in fe.items():                      # at the end of the arena: their gradients are the last to complete
            offs[n] = (o, prm.numel(), tuple(prm.shape))
            o = _up4(o + prm.numel())
        self.n_dense = o
        self.offs = offs
        # all-reduce buckets (float ranges of the arena): vocabulary projection and embedding table as soon as the backward has
        # them, then ONE bucket for everything that is complete only when the backward ends -- the hypernet's small layers and
        # second-layer biases (= d theta), feature_fc, attention, init_h and the domain front-end
        s_fc = offs["captioner.fc.weight"][0]
        e_fc = offs["captioner.embed.weight"][0]
        e_emb = _up4(e_fc + offs["captioner.embed.weight"][1])
        fe_lo = min([offs[n][0] for n in fe], default=o)
        assert e_emb == fe_lo or not fe
        self._buckets = {"tail": (0, s_fc), "fc": (s_fc, e_fc), "embed": (e_fc, e_emb), "frontend": (e_emb, o)}
        dev = self.dev
        self.flat_p = torch.zeros(o, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(o, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(o, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(o, dtype=torch.float32, device=dev)
        self._owned: Dict[str, torch.nn.Parameter] = {}
        for n, p in order:
            self._adopt(n, p)
        for n in self._dec_names:
            self._adopt("captioner." + n, dec[n])
        for n, prm in fe.items():
            self._adopt(n, prm)
        # rank-1 group
        self.W2 = [hyper[f"hn_heads.{i}.2.weight"] for i in range(nh)]
        for w in self.W2:
            w.data = w.data.contiguous()
        self.W2_m = [torch.zeros_like(w.data) for w in self.W2]
        self.W2_v = [torch.zeros_like(w.data) for w in self.W2]

    def _frontend_tensors(self):
        if self.frontend is None:
            return {}
        return {"embed." + n: prm for n, prm in self.frontend.named_parameters()}

    @classmethod
    def from_cc(cls, net_cc, **kw):
        """Trainer for a cc_train_hypernet.HyperNetCC: its hypernet, and its domain front-end unless the mode is 'one hot'."""
        if net_cc.embedding == "one hot":
            return cls(net_cc.hypernet, **kw)
        return cls(net_cc.hypernet, frontend=net_cc.embed, frontend_mode=net_cc.embedding, **kw)

    def _view(self, flat, name):
        cache = self._vcache.setdefault(id(flat), {})
        v = cache.get(name)
        if v is None:
            o, n, shape = self.offs[name]
            v = cache[name] = flat[o:o + n].view(shape)
        return v

    def _invalidate_caches(self):
        """The per-step dicts / ctypes structs are memoised (host time per step matters: a slow or contended host
        starves the GPU); anything that moves a parameter's storage clears them."""
        self._dt_cache = {}
        self._hp_cache = None
        self._hg_cache = None

    def _adopt(self, name, p):
        self._invalidate_caches()
        v = self._view(self.flat_p, name)
        v.copy_(p.data)
        p.data = v
        p.grad = self._view(self.flat_g, name)
        self._owned[name] = p

    def _sync_params(self):
        """Callers may swap sub-modules (captioner.embed = ...from_pretrained, hypernet_attention.py:108;
        sub-module transplant :424-428).  Re-adopt any parameter whose storage left the arena."""
        hyper = self.net.hyper_named_tensors()
        dec = {"captioner." + n: t for n, t in self.cap._named_tensors().items() if n in self._dec_names}
        if self.frontend is not None:
            dec.update(self._frontend_tensors())
        self._readopted = False
        vers = self._versions
        for name in self.offs:
            cur = hyper.get(name, dec.get(name))
            o, n, shape = self.offs[name]
            if cur.data_ptr() != self.flat_p.data_ptr() + 4 * o or tuple(cur.shape) != shape:
                if tuple(cur.shape) != shape:
                    raise CaphnError(f"{name}: shape changed to {tuple(cur.shape)}, arena holds {shape}")
                self._adopt(name, cur)
                self._readopted = True
            # torch-side writes into a parameter (load_state_dict, .copy_, an initialiser) bump its version counter;
            # libcaphn's own updates go through raw pointers and do not.  A bump means anything derived from the old
            # values -- the prefetched next theta, the side-stream precompute -- is stale.
            v = cur._version
            if vers.get(name) != v:
                if name in vers:
                    self._readopted = True
                vers[name] = v
        for i in range(self._nh):
            w = hyper[f"hn_heads.{i}.2.weight"]
            if w is not self.W2[i] or not w.data.is_contiguous():
                w.data = w.data.contiguous()
                self.W2[i] = w
                self._readopted = True
                self._invalidate_caches()
            v = w._version
            if vers.get(i) != v:
                if i in vers:
                    self._readopted = True
                vers[i] = v

    # ------------------------------------------------------------------ per-shape buffers
    def _buffers(self, B, T, P):
        key = (B, T, P)
        b = self._bufs.get(key)
        if b is None:
            import dataclasses
            # gz: the gradient arena is cleared once per step (forward_backward), not tensor by tensor
            dims = dataclasses.replace(self.cap.dec_dims(B, T, P), rows=self.skip_ignored_rows, gz=True)
            # the logits / d logits buffer is the trainer's own: its row pitch is padded to whole 128-byte lines (V = 9684: every
            # 128-byte store of the logits GEMM straddled two lines, every d logits row of the two backward GEMMs started mid-line;
            # tools/microbench_logits_pitch.py: logits -5 us, dHs -12 us, dW_fc -5 us)
            ld = (dims.V + 31) // 32 * 32 if self.pad_logits else dims.V
            if ld != dims.V:
                dims = dataclasses.replace(dims, logits_ld=ld)
            b = {"dims": dims, "ws": ops.decoder_workspace(dims, self.dev),
                 "logits": torch.empty(B, T, ld, dtype=torch.float32, device=self.dev),
                 "alphas": torch.empty(B, T, P, dtype=torch.float32, device=self.dev),
                 "loss": torch.zeros(2, dtype=torch.float32, device=self.dev),
                 "ce_ws": ops.ce_workspace(B * T, self.dev)}
            b["cnt_ptr"] = ops.decoder_rowcount_ptr(dims, b["ws"]) if dims.rows else None
            self._bufs[key] = b
        return b

    def _dec_tensors(self, theta_flat, grads: bool):
        """decoder parameter (or gradient) dict; the GRU entries are slices of theta / dtheta."""
        key = (theta_flat.data_ptr(), bool(grads))
        hit = self._dt_cache.get(key)
        if hit is not None:
            return hit
        flat = self.flat_g if grads else self.flat_p
        d = {n: self._view(flat, "captioner." + n) for n in self._dec_names}
        shapes = self.cap.dec_dims(1, 1, 1).param_shapes()
        o = 0
        for n in self._cell_names:
            k = 1
            for s in shapes[n]:
                k *= s
            d[n] = theta_flat[o:o + k].view(shapes[n])
            o += k
        assert o == self.theta_size, "hypernet heads do not match the cell's parameter sizes"
        d["__frozen__"] = True              # same dict object <-> same storage: ops memoise the ctypes struct in it
        d["__theta__"] = theta_flat         # keeps the storage (and therefore the key) alive
        self._dt_cache[key] = d
        return d

    def _hyper_params(self):
        hp = self._hp_cache
        if hp is None:
            hp = {n: self._owned[n].data for n in self.shape.param_names() if n in self._owned}
            for i in range(self._nh):
                hp[f"hn_heads.{i}.2.weight"] = self.W2[i].data
            hp["__frozen__"] = True
            self._hp_cache = hp
        return hp

    # ------------------------------------------------------------------ the step
    # ------------------------------------------------------------------ domain front-end (cc_train_hypernet.py:93-109, :136-149)
    def _fe_forward(self, inp, slot):
        """The hypernet's input row from `inp` (an index for 'embedding', the domain's feature vector otherwise); keeps what
        the backward needs in `slot`."""
        if self.fe_mode == "embedding":
            slot["idx"] = int(inp)
            return self._view(self.flat_p, "embed.weight")[int(inp)]
        inp = inp.reshape(1, -1).to(device=self.dev, dtype=torch.float32).contiguous()
        slot["inp"] = inp
        y0 = slot.get("y0")
        W0, b0 = self._view(self.flat_p, "embed.0.weight"), self._view(self.flat_p, "embed.0.bias")
        if y0 is None:
            y0 = slot["y0"] = torch.empty(1, W0.shape[0], dtype=torch.float32, device=self.dev)
        ops.gemm(inp, W0, tb=True, bias=b0, lrelu=True, out=y0)
        if self.fe_mode == "JSD":
            return y0.view(-1)
        W2, b2 = self._view(self.flat_p, "embed.2.weight"), self._view(self.flat_p, "embed.2.bias")
        x = slot.get("x")
        if x is None:
            x = slot["x"] = torch.empty(1, W2.shape[0], dtype=torch.float32, device=self.dev)
        ops.gemm(y0, W2, tb=True, bias=b2, lrelu=True, out=x)
        return x.view(-1)

    def _fe_backward(self, gx, slot):
        """d loss / d (front-end parameters) from the hypernet's input-row gradient gx, into the (zeroed) gradient arena."""
        g = lambda n: self._view(self.flat_g, n)
        if self.fe_mode == "embedding":
            i = slot["idx"]
            tok = self._toks.get(("emb", i))
            if tok is None:
                tok = self._toks[("emb", i)] = torch.full((1,), i, dtype=torch.int64, device=self.dev)
            ops.embedding_scatter_add(gx.view(1, -1), tok, g("embed.weight"))
            return
        y0 = slot["y0"]
        if self.fe_mode == "JSD":
            dz0 = ops.lrelu_bwd(gx.view(1, -1).contiguous(), y0)
        else:
            dz2 = ops.lrelu_bwd(gx.view(1, -1).contiguous(), slot["x"])
            ops.outer(dz2.view(-1), y0.view(-1), out=g("embed.2.weight"))
            ops.axpy_(g("embed.2.bias"), dz2.view(-1))
            dy0 = ops.gemm(dz2, self._view(self.flat_p, "embed.2.weight"))
            dz0 = ops.lrelu_bwd(dy0, y0)
        ops.outer(dz0.view(-1), slot["inp"].view(-1), out=g("embed.0.weight"))
        ops.axpy_(g("embed.0.bias"), dz0.view(-1))

    def _fe_key(self, inp):
        return ("emb", int(inp)) if self.fe_mode == "embedding" else ("fe", inp.data_ptr(), inp._version)

    def forward_backward(self, features, captions, x_style=None, style_token: Optional[int] = None,
                         validate: bool = False, domain_input=None, defer_loss: bool = False):
        """Fills the gradient arena (and the rank-1 factors) for one minibatch; returns the device
        tensor [loss, n_valid_targets] (a per-shape buffer the next call overwrites: .clone() or .item() it to keep a
        step's value).  Exactly one of x_style ([he] or [1,he]) / style_token (Flickr
        path: x = captioner.embed.weight[token], hypernet_attention.py:139-142)."""
        if (x_style is not None) + (style_token is not None) + (domain_input is not None) != 1:
            raise CaphnError("pass exactly one of x_style / style_token / domain_input")
        if domain_input is not None and self.frontend is None:
            raise CaphnError("domain_input needs a trainer built with a front-end (FusedTrainer.from_cc)")
        self._sync_params()
        if self._works:                 # a previous forward_backward whose optimizer_step never came: its collectives
            dp.wait_all(self._works)    # still own the gradient arena
            self._works = []
        B, P, _ = features.shape
        T = captions.shape[1]
        buf = self._buffers(B, T, P)
        dims = buf["dims"]
        if self.cap.training and self.cap.drop.p > 0:
            # h = self.drop(h) (models/decoderlstm.py:104): a fresh counter-based mask per step, the same seed in the backward
            import dataclasses
            from .functional import next_seed
            dims = dataclasses.replace(dims, drop_p=float(self.cap.drop.p), seed=next_seed())
            self.last_dropout_seed = dims.seed
        hp = self._hyper_params()
        if domain_input is not None:
            key = self._fe_key(domain_input)
            x = None                    # computed below unless the previous optimiser pass already did
        elif style_token is not None:
            x = self._view(self.flat_p, "captioner.embed.weight")[style_token]
        else:
            x = x_style.reshape(-1).to(device=self.dev, dtype=torch.float32)
        # every split-K / atomic target of the backward lives in this arena.  When the previous optimiser pass announced this
        # minibatch, everything behind d theta was cleared on the side stream beside the rank-1 Adam passes (which still read
        # d theta as their row factor): only that head of the arena is cleared here
        zero_lo_only = bool(getattr(self, "_zero_hi_done", False))
        self._zero_hi_done = False
        joined, self._aux_joined = self._aux_joined and self._pre_key is not None, False
        if zero_lo_only or self._aux_pending:   # this stream must not run ahead of the side clear (nothing has written the arena since)
            torch.cuda.current_stream().wait_event(self._lookup_done)      # / of the caption-side work in the workspace (joined:
        self._aux_pending = False                                          # that event also covers the precompute stream)
        theta = getattr(self, "_theta", None)
        if theta is None:
            theta = self._theta = torch.empty(self.theta_size, dtype=torch.float32, device=self.dev)
        if domain_input is None:
            key = ("tok", int(style_token)) if style_token is not None else ("x", x_style.data_ptr(), x_style._version)
        if self._next_key is not None and self._next_key == key and not self._readopted:
            # theta for this input was already produced by the previous optimiser pass
            self._theta, self._theta_next = self._theta_next, self._theta
            self._acts, self._acts_next = self._acts_next, self._acts
            self._fe_slot, self._fe_slot_next = self._fe_slot_next, self._fe_slot
            theta = self._theta
        else:
            if domain_input is not None:
                x = self._fe_forward(domain_input, self._fe_slot)
            ops.hyper_forward(self.shape, hp, x, theta=theta, acts=self._acts)
        self._next_key = None
        params = self._dec_tensors(theta, grads=False)
        fdims = dims
        rows_done = False
        pre = 0
        looked_up = False
        if self._pre_key is not None:
            kf, kc, kB, kT, kP, level = self._pre_key
            if (kf, kB, kT, kP) == (_tkey(features), B, T, P) and not self._readopted:
                pre = 7 if (level >= 2 and kc == _tkey(captions) and theta is self._theta_pre) else 1
                # level 1 with the captions announced: the live-row map and the embedding lookup are in the workspace already
                looked_up = pre == 1 and kc is not None and kc == _tkey(captions)
            self._pre_key = None
            self._pre_hold = None
            if pre == 1 and validate is False:
                # what does not depend on the side stream's work goes first, so it hides behind the wait
                if looked_up:
                    # ... except the lookup itself, which a side stream did (waited for above: the gate GEMM reads its rows)
                    rows_done = True
                    d64 = buf.get("d64")
                    if d64 is None or dims is not buf["dims"]:
                        import dataclasses as _dc
                        d64 = _dc.replace(dims, pre=64)
                        if dims is buf["dims"]:
                            buf["d64"] = d64
                    ops.decoder_inputs(d64, params, captions, buf["ws"])
                else:
                    if dims.rows:
                        ops.decoder_prepare_rows(dims, captions, 0, buf["ws"])
                        rows_done = True
                    ops.decoder_inputs(dims, params, captions, buf["ws"])
                pre = 5
                if level != 3 and self._pre_token is not None and ops.precompute_epoch(self.dev) == self._pre_token:
                    pre |= 16     # bit 16: the forward composite waits for the side stream's precompute itself, piecewise
                                  # (feature_fc output before G, the rest before the recurrent kernel).  The library keeps those
                                  # events per DEVICE: only valid while nobody has issued another precompute since ours
            if not (pre & 16) and not joined:
                torch.cuda.current_stream().wait_event(self._pre_done)      # even if the announced batch did not come
            if pre == 7:
                rows_done = True
        zero_lo_side = False
        lo_done, self._zero_lo_by_adam = self._zero_lo_by_adam, False
        if lo_done:                    # the rank-1 passes of the last optimiser step cleared d theta behind their reads
            if not zero_lo_only:
                ops.zero_(self.flat_g[_up4(self.theta_size):])
        elif zero_lo_only:
            # d theta was the rank-1 passes' row factor until a moment ago and is written again by the backward's weight-gradient
            # GEMMs, half a millisecond from now: its clear goes to the caption-side stream, behind this point of the caller's
            # stream, and is joined in front of the backward (a dependency satisfied that early costs nothing; in line it was a
            # launch between the last Adam pass and the recurrent kernel)
            aux = self._aux_stream
            if self._main_ev is None:
                self._main_ev, self._zero_lo_done = torch.cuda.Event(), torch.cuda.Event()
            self._main_ev.record()
            aux.wait_event(self._main_ev)
            with torch.cuda.stream(aux):
                ops.zero_(self.flat_g[:_up4(self.theta_size)])
                self._zero_lo_done.record(aux)
            zero_lo_side = True
        else:
            ops.zero_(self.flat_g)
        # bit 8 on both sides of the step: the forward zero-fills the backward's d Hs accumulator inside a kernel it launches
        # anyway, the backward skips its own fill (it sat on the chain between the loss and the first backward GEMM)
        dv = buf.get("dims_variants") if dims is buf["dims"] else None
        if dv is None:
            import dataclasses
            # bit 8 on both sides of the step: d Hs zero-filled by the forward's prep kernel; bit 32: ctx left by the forward (the
            # alternative -- ctx_in_forward = False: the backward forms it on a branch of its own beside BPTT -- measured equal
            # to 7 us slower, profiles/r03_step_boundary_ab.txt)
            dv = {p: dataclasses.replace(dims, pre=p | 8 | (32 if self.ctx_in_forward else 0)) for p in (0, 1, 5, 7, 21, 7 | 128)}
            if dims is buf["dims"]:          # (with decoder dropout the dims carry a fresh seed every step: not cached)
                buf["dims_variants"] = dv
        packed, self._pair_packed = self._pair_packed, None
        if pre == 7 and packed is not None:
            pk = ops.decoder_pair_pack_desc(dims, buf["ws"])
            if pk is not None and packed == (pk.wp, theta.data_ptr()):
                pre = 7 | 128        # the recurrent kernels' prep launch was issued beside the optimiser, W_hh packed by its pass
        fdims = dv[pre]
        if dims.rows and not rows_done:
            ops.decoder_prepare_rows(dims, captions, 0, buf["ws"])
        ops.decoder_forward(fdims, params, features, captions, buf["ws"], logits=buf["logits"], want_alphas=False,
                            validate=validate)
        # loss: d logits now; the reduction of the per-row losses to the reported scalar runs on the side stream (nothing
        # on the device waits for it; step() joins it at the end)
        cnt = buf["cnt_ptr"] if dims.rows else None
        ops.cross_entropy_rows(buf["logits"], captions, 0, buf["logits"], buf["ce_ws"], leave_ignored_rows=dims.rows,
                               n_valid_ptr=cnt, V=dims.V)
        dlogits = buf["logits"]
        # (the reduction of the per-row losses used to run on the side stream; the two events that took -- a record on
        #  this stream, a wait at the end of the backward -- cost the chain more than the 6 us kernel does in line)
        # (step(): the reduction rides in the optimiser's first launch instead -- nothing on the device needs the scalar)
        if defer_loss:
            self._ce_pending = (B * T, buf["ce_ws"], cnt, buf["loss"])
        else:
            ops.cross_entropy_finish(B * T, buf["ce_ws"], buf["loss"], cnt)
        dtheta = self.flat_g[:self.theta_size]
        grads = self._dec_tensors(dtheta, grads=True)
        hg = self._hg_cache
        if hg is None:
            hg = self._hg_cache = {n: self._view(self.flat_g, n) for n in self.offs if n.startswith("hn_")}
            hg["__frozen__"] = True
        if self._hyper_ws is None:
            import ctypes as C
            from . import _lib as L
            d = ops._hyper_desc(self.shape, hp)
            self._hyper_ws = torch.empty(L.load().caphn_hyper_backward_workspace_bytes(C.byref(d)),
                                         dtype=torch.uint8, device=self.dev)
        # decoder backward with the hypernet VJP hooked in: it starts on a side stream as soon as dtheta is
        # complete and streams the 576 MB of second-layer weights beside the attention / feature_fc chain
        # Flickr path on one rank: the style row's VJP is added to its row of the embedding gradient by the VJP's last kernel
        # (data parallel: the row goes to the communication stream, in front of the embedding bucket's all-reduce)
        direct = style_token is not None and not dp.active(self.group)
        if zero_lo_side:
            torch.cuda.current_stream().wait_event(self._zero_lo_done)
        gx = ops.decoder_hyper_backward(dv[0], params, features, captions, dlogits, grads, buf["ws"],
                                        self.shape, hp, self._acts, hg, self._hyper_ws,
                                        want_x=(style_token is not None or domain_input is not None) and not direct,
                                        x_accum_into=self._view(self.flat_g, "captioner.embed.weight")[int(style_token)] if direct else None)
        if domain_input is not None:    # the input row's gradient goes on through the front-end, before the exchange is issued
            self._fe_backward(gx, self._fe_slot)
        tok = None
        if style_token is not None:
            # Flickr path: the style row of the embedding also feeds the hypernet -- its VJP is added to the
            # embedding gradient before that gradient is reduced
            tok = self._toks.get(int(style_token))
            if tok is None:
                tok = self._toks[int(style_token)] = torch.full((1,), int(style_token), dtype=torch.int64, device=self.dev)
        if dp.active(self.group):
            self._issue_exchange(gx, tok)
        if self._loss_pending:       # whoever reads the returned loss on the current stream sees the finished value
            torch.cuda.current_stream().wait_event(self._loss_done)
            self._loss_pending = False
        return buf["loss"]

    def _issue_exchange(self, gx, tok):
        """Data-parallel exchange, started while the backward is still running: the collectives are issued
        from a side stream that waits for the backward's milestones (caphn_decoder_backward_milestone), in the order the
        backward reaches them; RCCL runs them on its own stream beside the attention / feature_fc chain and the hypernet's
        576 MB transposed GEMV.  Nothing is allocated or packed per step: dtheta is gathered straight out of the gradient
        arena, the activation factors out of the acts buffer, into two preallocated [R, .] buffers.
          after dL/dtheta          ONE all-gather of the packed rank-1 factors [d theta | acts] (a SNAPSHOT of dtheta: the arena
                                   range itself is all-reduced later)
          after the vocab wgrad    all-reduce of fc.weight / fc.bias                       (7.8 MB)
          after the scatter-add    all-reduce of embed.weight (+ the style row's VJP, Flickr path)   (7.7 MB)
          after the whole call     ONE all-reduce of [second-layer biases | hn_base | first layers | feature_fc, attention,
                                   init_h] (3.9 MB; + the domain front-end when there is one): these are what the two chains
                                   behind BPTT finish last, within ~20 us of each other -- no collective is issued after that"""
        R = dp.world(self.group)
        g, th = self.flat_g, self.theta_size
        if self._gfac_all is None or self._gfac_all.shape[0] != R:
            # ONE packed factor buffer per rank: [d theta (theta_size, padded to 4) | acts]; the gathered [R, .] buffer is viewed as
            # the row factors and the column factors (row pitch = the packed length: the rank-R kernels take leading dimensions)
            L = self._acts.numel()
            tp = _up4(th)
            # (torch.empty, not zeros: a fill kernel would run on the CALLER's stream, unordered against the snapshot copies the
            #  communication stream makes below -- on the first step it could land after them and wipe the factors; the pad
            #  columns are never read)
            self._fac_snap = torch.empty(_up4(tp + L), dtype=torch.float32, device=self.dev)
            self._fac_all = torch.empty(R, _up4(tp + L), dtype=torch.float32, device=self.dev)
            self._gfac_all = self._fac_all[:, :th]
            self._acts_all = self._fac_all[:, tp:tp + L]
        main = torch.cuda.current_stream()
        cs = self._comm_stream
        bk = self._buckets
        w = self._works
        if getattr(self, "time_collectives", False):
            # measurement mode (bench.py, a few steps OUTSIDE the timed region): every collective is waited for at once between
            # two events on the communication stream, so each one's own device time is seen; nothing overlaps in this mode
            class _Timed(list):
                def append(inner, work):
                    if work is not None:
                        work.wait()
                    e = torch.cuda.Event(enable_timing=True); e.record(cs)
                    self._xstat.setdefault("marks", []).append(e)
            w = _Timed()
            with torch.cuda.stream(cs):
                cs.wait_stream(main)
                e = torch.cuda.Event(enable_timing=True); e.record(cs)
                self._xstat.setdefault("marks", []).append(e)
        with torch.cuda.stream(cs):
            ops.backward_milestone_wait(ops.MS_DTHETA)
            # the all-gather reads a snapshot: dtheta's arena range is part of the "tail" bucket, all-reduced in place below, and a
            # backend that runs asynchronous works out of issue order (gloo's thread pool) must not see the two overlap
            ops.stream_copy(g[:th], self._fac_snap[:th])
            ops.stream_copy(self._acts, self._fac_snap[_up4(th):_up4(th) + self._acts.numel()])
            w.append(dp.all_gather_factors(self._fac_snap, self._fac_all, self.group, async_op=True)[1])
            ops.backward_milestone_wait(ops.MS_VOCAB)
            w.append(dp.all_reduce_dense(g[bk["fc"][0]:bk["fc"][1]], self.group, async_op=True))
            ops.backward_milestone_wait(ops.MS_EMBED)
            if tok is not None:
                ops.backward_milestone_wait(ops.MS_HYPER)      # gx is the last output of the hypernet VJP
                gx.record_stream(cs)        # allocated on the main stream, consumed here
                ops.embedding_scatter_add(gx.view(1, -1), tok, self._view(g, "captioner.embed.weight"))
            w.append(dp.all_reduce_dense(g[bk["embed"][0]:bk["embed"][1]], self.group, async_op=True))
            cs.wait_stream(main)
            w.append(dp.all_reduce_dense(g[bk["tail"][0]:bk["tail"][1]], self.group, async_op=True))
            if bk["frontend"][1] > bk["frontend"][0]:
                w.append(dp.all_reduce_dense(g[bk["frontend"][0]:bk["frontend"][1]], self.group, async_op=True))

    def _exchange(self):
        """Joins the exchange _issue_exchange started.  Returns (gfac [R,theta], acts_all [R,L])."""
        if not dp.active(self.group):
            return self.flat_g[:self.theta_size].view(1, -1), self._acts.view(1, -1)
        xs = self._xstat
        xs["steps"] += 1
        xs["collectives"] += len(self._works)
        timed = xs["steps"] % 8 == 0 and len(xs["timed"]) < 64       # event pairs cost queue time: every 8th step only
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dp.wait_all(self._works)
        torch.cuda.current_stream().wait_stream(self._comm_stream)
        if timed:
            e1.record()
            xs["timed"].append((e0, e1))
        self._works = []
        return self._gfac_all, self._acts_all

    def collective_report(self) -> dict:
        """What the data-parallel exchange did so far: backend, world size, collectives issued per step, and the time the main
        stream spent waiting for them in front of the optimiser (HIP events around the join; the part of the exchange that
        the backward did not hide)."""
        import torch.distributed as dist
        if not dp.active(self.group):
            return {"backend": None, "world": 1, "per_step": 0, "exposed_us": 0.0}
        xs = self._xstat
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) * 1e3 for a, b in xs["timed"]]
        per = None
        marks = xs.get("marks")
        if marks:
            names = ["all_gather factors", "all_reduce fc", "all_reduce embed", "all_reduce tail", "all_reduce frontend"]
            nper = 5 if self._buckets["frontend"][1] > self._buckets["frontend"][0] else 4
            steps = [marks[i:i + nper + 1] for i in range(0, len(marks) - nper, nper + 1)]
            per = {names[j]: float(sum(st[j].elapsed_time(st[j + 1]) for st in steps) / len(steps) * 1e3) for j in range(nper)}
        return {"backend": dist.get_backend(self.group), "world": dp.world(self.group),
                "per_step": xs["collectives"] / max(1, xs["steps"]), "per_collective_us_serialised": per,
                "exposed_us": (sum(ts) / len(ts)) if ts else None, "exposed_us_max": max(ts) if ts else None,
                "steps_observed": xs["steps"], "steps_timed": len(ts)}

    def _begin_step(self):
        self.step_count += 1
        a, b = ops.adam_scalars(self.lr, self.betas, self.step_count)
        if self._graph_scalars:
            i = self.step_count % 64
            ev = self._adam_copied[i]
            if ev is not None:
                ev.synchronize()         # 64 steps ago: the host only blocks here if it runs that far ahead of the GPU
            slot = self._adam_host[i]
            slot[0] = a
            slot[1] = b
            self._adam_dev.copy_(slot, non_blocking=True)
            if ev is None:
                ev = self._adam_copied[i] = torch.cuda.Event()
            ev.record()

    def _optimizer_impl(self, next_x_style=None, next_style_token=None, next_batch=None, next_domain_input=None):
        R = dp.world(self.group)
        gfac, acts_all = self._exchange()
        o = 0
        segs = []
        for i, (k, w) in enumerate(self.shape.heads):
            ao, an = self._acts_layout[f"a{i}"]
            gi, ai = gfac[:, o:o + w], acts_all[:, ao:ao + an]
            segs.append((gi, ai, o, w, ao, an))
            o += w
        # global gradient norm (dense arena + Gram-matrix norm of the rank-R second-layer gradients) and the clip
        # coefficient, one launch
        gn = self._gradnorm.get(R)
        if gn is None:
            gn = self._gradnorm[R] = ops.GradNorm(self.n_dense, R, len(segs), self.dev)
        step = max(self.step_count, 1)
        ce, self._ce_pending = getattr(self, "_ce_pending", None), None
        # norm partials, then ONE launch that finishes the clip coefficient, runs Adam over the dense arena and reduces the loss
        ops.grad_norm_adam_dense(gn, self.flat_p, self.flat_m, self.flat_v, self.flat_g, [(s[0], s[1]) for s in segs], self.max_norm,
                                 1.0 / R, self._coef, self.lr, step, self.betas, self.eps,
                                 dev_scalars=self._adam_dev if self._graph_scalars else None, ce=ce)
        prefetch = (next_x_style is not None) or (next_style_token is not None) or (next_domain_input is not None)
        known = (prefetch and next_batch is not None and next_batch[1] is not None and len(segs) == 4 and
                 next_batch[1].dtype == torch.int64 and next_batch[1].is_contiguous())
        full = self.overlap_level == 2 and known
        split = self.overlap_level >= 3 and known and self.overlap_after_head == 0
        # level 4: the theta-independent front does not wait for the W_ih pass at all -- it is forked HERE, behind the dense Adam
        # launch and in front of the three tiny GEMVs of the next activations, so that its first GEMM has its workgroups on the chip
        # before the W_ih pass arrives (ten microseconds of head start: 84 us instead of 256 for feature_fc.0 beside a rank-1 pass);
        # beside the LAST pass only the two GEMMs that need theta are left (G, x-side gates: done long before the pass ends)
        early = split and self.overlap_level >= 4
        pack = None

        def early_fork():
            nonlocal early, pack
            self._precompute_next(*next_batch, level=1, lookup_dp=True)
            early = self._pre_key is not None and self._pre_key[1] is not None     # (the captions were looked up)
            if early and self.pack_in_adam:
                # the pair recurrent kernels' prep launch (exchange areas, d Hs, the U_a rows of the packed weight copy) goes to the
                # caption-side stream now; the W_hh rows are written by the W_hh pass itself: nothing is left between the last
                # rank-1 pass and the next recurrent kernel
                nf, nc, nT = next_batch
                nbuf = self._buffers(nf.shape[0], nT, nf.shape[1])
                pack = ops.decoder_pair_pack_desc(nbuf["dims"], nbuf["ws"])
                if pack is not None:
                    with torch.cuda.stream(self._aux_stream):
                        ops.decoder_pair_prep(nbuf["dims"], self._dec_tensors(self._theta, grads=False), nbuf["ws"])
                        self._lookup_done.record(self._aux_stream)
        if early and not self.fork_after_acts:
            early_fork()
        if prefetch:
            # the small layers (and the style row of the embedding / the front-end) are already updated: compute the next
            # step's head activations, then let the rank-1 Adam pass emit theta_next = W2' a' + b2' row by row
            hp = self._hyper_params()
            if next_domain_input is not None:
                xn = self._fe_forward(next_domain_input, self._fe_slot_next)
                self._next_key = self._fe_key(next_domain_input)
                self._next_hold = next_domain_input
            elif next_style_token is not None:
                xn = self._view(self.flat_p, "captioner.embed.weight")[int(next_style_token)]
                self._next_key = ("tok", int(next_style_token))
            else:
                xn = next_x_style.reshape(-1).to(device=self.dev, dtype=torch.float32)
                self._next_key = ("x", next_x_style.data_ptr(), next_x_style._version)
                self._next_hold = next_x_style        # the announced tensor stays alive: its address cannot be recycled
            ops.hyper_forward_acts(self.shape, hp, xn, self._acts_next)
            if self._theta_next is None:
                self._theta_next = torch.empty(self.theta_size, dtype=torch.float32, device=self.dev)
        if early and self.fork_after_acts:
            early_fork()        # (experiment: ONE record on the caller's stream -- behind the next activations -- serves the front and the bias heads)
        # Order of the rank-1 passes when the next minibatch is known: everything the front of the next forward needs
        # (W_ih, b_ih -- heads 0, 2; and the tiny b_hh head) first, then the side stream starts the next forward's
        # front end while the W_hh pass (head 1, HBM-bound) is still streaming.
        order = [0, 2, 3, 1] if (full or split) else list(range(len(segs)))
        # (measured and dropped: W_hh and the bias passes FIRST with the front beside them and the W_ih pass last and alone -- G and the
        #  gate GEMM then sit between the last pass and the recurrent kernel: 1.636/1.661 against 1.585/1.600 ms, same box)
        fork_after = order[-2] if full else self.overlap_after_head
        if next_batch is not None and not full and not early and self.overlap_after_head < 0:
            self._precompute_next(*next_batch, level=1)
        # consecutive small heads (the bias heads: [3H, k]) go out as ONE launch -- each is ~11 us of launch-bound kernel
        groups = []
        for i in order:
            small = segs[i][3] <= 16384
            if groups and small and groups[-1][1]:
                groups[-1][0].append(i)
            else:
                groups.append(([i], small))
        dev_sc = self._adam_dev if self._graph_scalars else None
        # one rank, the row factor is d theta in the arena itself: every pass clears its rows behind the read (it is their last
        # reader), so the next forward has no fill of d theta to launch in front of its recurrent kernel
        zg = self.zero_by_adam and R == 1 and not dp.active(self.group) and self.theta_size % 4 == 0

        def jobs_of(members):
            out = []
            for i in members:
                gi, ai, o, w, ao, an = segs[i]
                job = [self.W2[i].data, self.W2_m[i], self.W2_v[i], gi, ai]
                if prefetch:
                    job += [self._acts_next[ao:ao + an], self._owned[f"hn_heads.{i}.2.bias"].data, self._theta_next[o:o + w]]
                out.append(job)
            return out
        if early and self.bias_heads_aside and len(groups) == 3 and groups[1][1] and not groups[0][1] and not groups[2][1]:
            # split front: the two bias heads' launch leaves the caller's stream -- it needs the clip coefficient and the next
            # activations only, so it runs on the caption-side stream beside the W_ih pass (b_ih is then ready long before the gate
            # GEMM behind that pass wants it), and the W_ih and W_hh passes follow each other directly
            if not self.fork_after_acts:     # (else: the caption-side stream already waits for a record behind the activations)
                if self._acts_ev is None:
                    self._acts_ev = torch.cuda.Event()
                self._acts_ev.record()
                self._aux_stream.wait_event(self._acts_ev)
            with torch.cuda.stream(self._aux_stream):
                ops.adam_rank_multi(jobs_of(groups[1][0]), self._coef, self.lr, step, self.betas, self.eps, dev_scalars=dev_sc, zero_gfac=zg)
                self._lookup_done.record(self._aux_stream)
            groups = [groups[0], groups[2]]
            order = [order[0], order[-1]]
        for members, _ in groups:
            jobs = jobs_of(members)
            if len(jobs) == 1 and pack is not None and members[0] == 1:
                ops.adam_rank_multi(jobs, self._coef, self.lr, step, self.betas, self.eps, dev_scalars=dev_sc, zero_gfac=zg, packs=[pack])
                self._pair_packed = (pack.wp, self._theta_next.data_ptr())
            elif len(jobs) == 1:
                j = jobs[0]
                kw = dict(next_a=j[5], next_bias=j[6], next_theta=j[7]) if prefetch else {}
                ops.adam_rank(j[0], j[1], j[2], j[3], j[4], self._coef, self.lr, step, self.betas, self.eps, dev_scalars=dev_sc,
                              zero_gfac=zg, **kw)
            else:
                ops.adam_rank_multi(jobs, self._coef, self.lr, step, self.betas, self.eps, dev_scalars=dev_sc, zero_gfac=zg)
            if fork_after in members and next_batch is not None and not early:
                # the side stream starts HERE on the device (an event), but its launches are enqueued after the remaining passes:
                # the host otherwise spends ~50 us on the precompute's launches before it gets to the next pass
                if self._fork_ev is None:
                    self._fork_ev = torch.cuda.Event()
                self._fork_ev.record()
                fork_level = 2 if full else 3 if split else 1
            if split and order[-2] in members:         # b_ih (and b_hh) exist: the x-side gate GEMM may go
                if self._bih_ev is None:
                    self._bih_ev = torch.cuda.Event()
                self._bih_ev.record()
            if fork_after in members and next_batch is not None and not (self.fork_late or split):
                self._precompute_next(*next_batch, level=fork_level)
        self._zero_lo_by_adam = zg
        if early:
            self._front_theta_next(*next_batch)
        elif next_batch is not None and fork_after >= 0 and fork_after in order and (self.fork_late or split):
            self._precompute_next(*next_batch, level=fork_level, after=self._fork_ev)
        return self._coef

    @staticmethod
    def _dims_pre(buf, dims, bits):
        """dims with the given dims.precomputed bits (cached per shape while the dims are the cached ones)."""
        import dataclasses
        if dims is not buf["dims"]:
            return dataclasses.replace(dims, pre=bits)
        c = buf.setdefault("dims_pre", {})
        d = c.get(bits)
        if d is None:
            d = c[bits] = dataclasses.replace(dims, pre=bits)
        return d

    def _front_theta_next(self, features, captions, T):
        """Second half of the split front (overlap_level 4; the first half is _precompute_next(level=1), issued before the rank-1
        passes): G on the precompute stream behind the W_ih pass, the x-side gate GEMM on the caption-side stream behind the b_ih
        pass.  Upgrades the announcement to level 3."""
        B, P, _ = features.shape
        buf = self._buffers(B, T, P)
        dims = buf["dims"]
        theta = self._theta_next
        params = self._dec_tensors(theta, grads=False)
        aux = self._aux_stream
        aux.wait_event(self._bih_ev)
        with torch.cuda.stream(aux):
            ops.decoder_inputs(self._dims_pre(buf, dims, 64), params, captions, buf["ws"])
        self._aux_pending = True
        self._pre_stream.wait_event(self._bih_ev)      # (one record behind the bias heads serves both: W_ih is older than that)
        with torch.cuda.stream(self._pre_stream):
            # (pre bits on this call: 1 = the theta-independent part is in the workspace already, 4 = the x side is the other stream's)
            ops.decoder_precompute(self._dims_pre(buf, dims, 1 | 4), params, features, buf["ws"], captions=captions)
            self._pre_done.record(self._pre_stream)
        # the caption-side stream joins the precompute stream (G and the gate GEMM ran side by side): the next forward then waits
        # for ONE event in front of its recurrent kernel -- every wait on the caller's stream is a packet between the last Adam
        # pass and that kernel
        aux.wait_event(self._pre_done)
        self._lookup_done.record(aux)
        self._aux_joined = True
        self._pre_token = ops.precompute_epoch(self.dev)
        self._theta_pre = theta
        kf, kc, kB, kT, kP, _ = self._pre_key
        self._pre_key = (kf, kc, kB, kT, kP, 3)

    def _precompute_next(self, features, captions, T, level=1, after=None, lookup_dp=False):
        """The decoder's dense parameters are final (adam_dense ran) and the workspace is free (backward is done):
        run the front of the next minibatch's forward on a side stream while the remaining Adam pass streams the
        hypernet (HBM-bound).  level 1: feature_fc / init_hidden / W_a f (theta-independent).  level 2 (next theta's
        W_ih / b_ih already produced by the passes issued so far, captions known): also the live-row map, G, the
        embedding lookup and the x-side gates, so the next forward starts at the recurrent kernel.  The fork sits
        after the head-0 pass, the kernel the roofline figure is quoted on, so that one is measured undisturbed."""
        B, P, _ = features.shape
        buf = self._buffers(B, T, P)
        dims = buf["dims"]
        theta = self._theta_next if level >= 2 else (self._theta if getattr(self, "_theta", None) is not None
                                                     else torch.empty(self.theta_size, device=self.dev))
        params = self._dec_tensors(theta, grads=False)
        main = torch.cuda.current_stream()
        if after is None:
            # ONE record on the caller's stream for both side streams (a record costs the recording stream ~6 us of queue time:
            # two of them sat between the dense Adam launch and the GEMVs of the next activations)
            if self._fork0_ev is None:
                self._fork0_ev = torch.cuda.Event()
            self._fork0_ev.record(main)
            after = self._fork0_ev
        self._pre_stream.wait_event(after)
        lookup = (level in (1, 3) and captions is not None and captions.dtype == torch.int64 and captions.is_contiguous()
                  and tuple(captions.shape) == (B, T) and (level == 3 or lookup_dp or not dp.active(self.group)))
        if level == 3 and not lookup:
            level = 1
        if lookup:
            # caption-only work of the next forward's front (live-row map, embedding lookup from the table adam_dense just
            # updated) and the clear of the gradient arena behind d theta (the rank-1 passes still read d theta itself): three
            # small launches on a stream of their own -- in front of the feature_fc GEMM on the precompute stream they delayed it,
            # and with it init_hidden / W_a f, which the next recurrent kernel waits for
            aux = getattr(self, "_aux_stream", None)
            if aux is None:
                aux = self._aux_stream = torch.cuda.Stream(device=self.dev)
                self._lookup_done = torch.cuda.Event()
            aux.wait_event(after)
            clear = not dp.active(self.group)
            with torch.cuda.stream(aux):
                if clear:
                    ops.zero_(self.flat_g[_up4(self.theta_size):])
                if dims.rows:
                    ops.decoder_prepare_rows(dims, captions, 0, buf["ws"])
                ops.decoder_lookup(dims, params, captions, buf["ws"])
                if level == 3:
                    # the x-side gate pre-activations Xe W_ih[:, :E]^T + b_ih of the NEXT theta: W_ih exists since the fork, b_ih
                    # since the event behind its pass
                    aux.wait_event(self._bih_ev)
                    ops.decoder_inputs(self._dims_pre(buf, dims, 64), params, captions, buf["ws"])
                self._lookup_done.record(aux)
            self._zero_hi_done = clear
            self._aux_pending = True
        with torch.cuda.stream(self._pre_stream):
            if level == 2:
                cl = captions
                if dims.rows:
                    ops.decoder_prepare_rows(dims, cl, 0, buf["ws"])
                ops.decoder_precompute(dims, params, features, buf["ws"], captions=cl)
            elif level == 3:     # theta-independent front, then G (pre bit 4 on this call: the x side is the other stream's)
                ops.decoder_precompute(self._dims_pre(buf, dims, 4), params, features, buf["ws"], captions=captions)
            else:
                ops.decoder_precompute(dims, params, features, buf["ws"])
            self._pre_done.record(self._pre_stream)
        self._pre_token = ops.precompute_epoch(self.dev)
        self._theta_pre = theta if level >= 2 else None
        self._pre_key = (_tkey(features), _tkey(captions) if ((level >= 2 or lookup) and captions is not None) else None, B, T, P, level)
        self._pre_hold = (features, captions)         # announced tensors stay alive: their addresses cannot be recycled

    def optimizer_step(self, next_x_style=None, next_style_token=None, next_batch=None, next_domain_input=None):
        """clip_grad_norm_(max_norm) over ALL gradients + Adam, on device, no host sync.
        If the NEXT minibatch's style row is already known (the data loader is one batch ahead), pass it:
        the Adam pass over the big second-layer weights then also produces the next step's theta, saving that
        step's 576 MB forward read (the following forward_backward must be called with that same input).
        On one rank the rank-1 passes CONSUME d theta (they clear it behind their read, zero_by_adam): grad() of the generated
        cell weights / w2_grad_dense() are meaningful between forward_backward and this call, not after it."""
        self._begin_step()
        return self._optimizer_impl(next_x_style, next_style_token, next_batch, next_domain_input)

    def step(self, features, captions, x_style=None, style_token=None, next_x_style=None, next_style_token=None,
             next_features=None, next_captions=None, next_T=None, domain_input=None, next_domain_input=None):
        """next_features (+ next_captions, or next_T when only the length is known; default: this T): the NEXT
        minibatch when the loader is one batch ahead -- the front of its forward then overlaps this step's optimiser
        (the next call must pass those same tensors; captions as int64)."""
        self._begin_step()          # the Adam scalars' H2D copy goes in front of the forward, off the optimiser's tail
        loss = self.forward_backward(features, captions, x_style, style_token, domain_input=domain_input, defer_loss=True)
        if next_captions is not None:
            next_T = next_captions.shape[1]
        nb = None if next_features is None else (next_features, next_captions, captions.shape[1] if next_T is None else next_T)
        self._optimizer_impl(next_x_style, next_style_token, nb, next_domain_input)
        return loss

    def step_graphed(self, features, captions, x_style=None, style_token=None):
        """Same step, replayed from a hipGraph (single GPU): ~90 kernel launches become one graph launch.
        Graphs are cached per (input buffers, shapes, style): the caller must keep feeding the same
        device buffers (copy new minibatches into them).  The first call with a key runs eagerly
        (creates the lazily allocated buffers), the second captures, later ones replay."""
        if dp.world(self.group) != 1:
            return self.step(features, captions, x_style, style_token)
        key = (features.data_ptr(), captions.data_ptr(), tuple(features.shape), tuple(captions.shape), style_token,
               None if x_style is None else x_style.data_ptr())
        if key not in self._seen:
            self._seen.add(key)
            return self.step(features, captions, x_style, style_token)
        # a replayed graph cannot take the step-dependent Adam scalars by value: they come from device memory,
        # refreshed (stream-ordered, from this step's pinned slot) in front of every replay
        self._graph_scalars = True
        try:
            self._begin_step()
            g = self._graphs.get(key)
            if g is None:
                self._sync_params()
                # nothing from outside the capture may be pending inside it: a side-stream precompute announced by an
                # earlier eager step (its event would be waited on by a capturing stream), a prefetched theta
                if self._pre_key is not None:
                    torch.cuda.current_stream().wait_event(self._pre_done)
                if getattr(self, "_zero_hi_done", False) or self._aux_pending:     # a side clear of the arena / caption-side work issued
                    torch.cuda.current_stream().wait_event(self._lookup_done)      # by an eager step: join it here; the captured step
                    self._zero_hi_done = self._aux_pending = False                 # clears the whole arena
                self._pre_key = self._pre_hold = self._next_key = self._next_hold = None
                torch.cuda.current_stream().wait_stream(self._pre_stream)
                self._zero_lo_by_adam = False       # the captured forward clears the WHOLE arena: a replay must not depend on what
                g = torch.cuda.CUDAGraph()          # ran before it (an eager forward_backward without its optimiser step)
                with torch.cuda.graph(g):
                    loss = self.forward_backward(features, captions, x_style, style_token, defer_loss=True)
                    self._optimizer_impl()
                self._graphs[key] = (g, loss)
            g, loss = self._graphs[key]
            g.replay()
        finally:
            self._graph_scalars = False
        return loss

    # ------------------------------------------------------------------ checkpoint / resume
    def optimizer_param_names(self) -> List[str]:
        """Parameter order of the reference's optimiser (hypernet_attention.py:124-130 / cc_train_hypernet.py:110-118):
        hn_heads, hn_base, captioner.feature_fc, .embed, .fc, .attention, .init_h (+ init_c for the LSTM configuration)."""
        names = []
        for i in range(self._nh):
            names += [f"hn_heads.{i}.0.weight", f"hn_heads.{i}.0.bias", f"hn_heads.{i}.2.weight", f"hn_heads.{i}.2.bias"]
        names += list(self._frontend_tensors())       # cc_train_hypernet.py:111-113: self.embed right after the heads
        names += ["hn_base.0.weight", "hn_base.0.bias", "hn_base.2.weight", "hn_base.2.bias"]
        dec = ["feature_fc.0.weight", "feature_fc.0.bias", "feature_fc.2.weight", "feature_fc.2.bias", "embed.weight",
               "fc.weight", "fc.bias", "attention.W_a.weight", "attention.W_a.bias", "attention.U_a.weight",
               "attention.U_a.bias", "attention.v_a.weight", "attention.v_a.bias", "init_h.weight", "init_h.bias",
               "init_c.weight", "init_c.bias"]
        names += ["captioner." + n for n in dec if n in self._dec_names]
        names += ["captioner." + n for n in self._dec_names if n.startswith("layers.")]      # (not in the reference's HyperNet list:
        return names                                                                          #  its captioner has one layer)

    def _moments(self, name):
        if name.startswith("hn_heads.") and name.endswith(".2.weight"):
            i = int(name.split(".")[1])
            return self.W2_m[i], self.W2_v[i]
        return self._view(self.flat_m, name), self._view(self.flat_v, name)

    def state_dict(self) -> dict:
        """Optimiser state in torch.optim.Adam's state_dict layout (what Lightning checkpoints for the reference,
        cc_train_hypernet.py:393): state[i] = {step, exp_avg, exp_avg_sq} with i the parameter's position in the reference's
        optimiser list, one param_group; `param_names` (extra key) spells that order out.  Parameters themselves are
        ordinary module state (net.state_dict())."""
        names = self.optimizer_param_names()
        state = {}
        for i, n in enumerate(names):
            m, v = self._moments(n)
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().clone(),
                        "exp_avg_sq": v.detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "param_names": names, "max_norm": self.max_norm}

    def load_state_dict(self, sd: dict) -> None:
        """Inverse of state_dict (also accepts a torch.optim.Adam state_dict built over the reference's parameter list).
        Call after the module's own load_state_dict."""
        names = self.optimizer_param_names()
        if "param_names" in sd and list(sd["param_names"]) != names:
            raise CaphnError("optimiser state was saved for a different parameter list")
        st = sd["state"]
        if len(st) not in (0, len(names)):
            raise CaphnError(f"optimiser state holds {len(st)} entries, this trainer has {len(names)} parameters")
        steps = set()
        for i, n in enumerate(names):
            if i not in st:
                continue
            m, v = self._moments(n)
            m.copy_(st[i]["exp_avg"].to(device=self.dev, dtype=torch.float32).view_as(m))
            v.copy_(st[i]["exp_avg_sq"].to(device=self.dev, dtype=torch.float32).view_as(v))
            steps.add(int(st[i]["step"]))
        if len(steps) > 1:
            raise CaphnError(f"per-parameter step counts differ ({sorted(steps)}): one fused step counter cannot represent them")
        self.step_count = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        if "max_norm" in sd:
            self.max_norm = sd["max_norm"]
        # whatever was derived from the previous state is stale
        self._next_key = self._next_hold = None
        if self._pre_key is not None:
            torch.cuda.current_stream().wait_event(self._pre_done)
        self._pre_key = self._pre_hold = None
        self._graphs.clear(); self._seen.clear()

    # ------------------------------------------------------------------ introspection for tests
    def grad(self, name):
        return self._view(self.flat_g, name)

    def w2_grad_dense(self, i):
        """Materialise dW2_i (tests only): sum_r dtheta_r (x) a_r of the LOCAL rank."""
        o = sum(w for _, w in self.shape.heads[:i])
        k, w = self.shape.heads[i]
        ao, an = self._acts_layout[f"a{i}"]
        return ops.outer(self.flat_g[o:o + w].contiguous(), self._acts[ao:ao + an].contiguous())


def _tkey(t):
    """Identity of an announced input: its address AND its version counter (an in-place refill of the same buffer, or a
    new tensor that the caching allocator placed at a freed address, is a different input)."""
    return (t.data_ptr(), t._version)


def L_sumsq_blocks(n):
    from . import _lib as L
    return L.load().caphn_sumsq_blocks(n)
