"""torch.optim.Adam (+ Lightning's gradient_clip_val) on libcaphn, as a real torch.optim.Optimizer.

This is what the module API's configure_optimizers() returns (hypernet_attention.py:123-133, cc_train_hypernet.py:110-122,
hypernet.py:116-123 build `torch.optim.Adam(params, lr)` and hand it to ReduceLROnPlateau; the Trainer clips the global norm
to 5.0, cc_train_hypernet.py:405), so an unchanged driver runs the optimiser -- the dominant HBM stream of the step -- in HIP:

  * `param_groups`, `state` (step / exp_avg / exp_avg_sq per parameter) and `state_dict()` are torch.optim.Adam's: LR schedulers
    wrap it, checkpoints interchange with torch.optim.Adam in both directions;
  * global-norm clipping happens INSIDE step() (`max_norm`): one launch computes clip_grad_norm_'s coefficient over every dense
    gradient and every rank-1 member, the update kernels multiply by it -- gradients are never rewritten;
  * parameters in separate allocations are updated by multi-tensor launches (caphn_adam_multi_f32), not one launch per tensor;
  * a hypernetwork's big second-layer weights W2_i [w_i, k_i] (576 MB of the reference configuration's 580) never get a dense
    gradient: adopt_rank1(hypernet) makes the backward hand over the factors (dtheta_i, a_i) and step() applies
    caphn_adam_rank_f32 (24 B/parameter, product formed on the fly); their share of the clip norm comes from
    ||g (x) a||^2 = (g.g)(a.a).
"""
from typing import Iterable, Optional

import torch

from . import ops
from ._lib import CaphnError


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: Optional[float] = None):
        """max_norm: torch.nn.utils.clip_grad_norm_(all parameters, max_norm) before the update (None: no clipping).  It is a
        plain attribute: a LightningModule's configure_gradient_clipping hook sets it from the Trainer's gradient_clip_val."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False)
        super().__init__(params, defaults)
        n = 0
        for group in self.param_groups:
            for p in group["params"]:
                n += 1
                if not p.is_cuda or p.dtype != torch.float32:
                    raise CaphnError("FusedAdam needs fp32 parameters on a CUDA(HIP) device (there is no CPU fallback)")
        if n == 0:
            raise ValueError("FusedAdam got no parameters")
        self.max_norm = max_norm
        dev = self.param_groups[0]["params"][0].device
        self._one = torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev)
        self._coef = torch.zeros(2, dtype=torch.float32, device=dev)
        self._norm_ws = None
        self._rank1 = {}         # id(parameter) -> (factor sink dict, key)
        self.grad_norm = None    # device tensor [clip coefficient, total norm] of the last clipped step

    # ------------------------------------------------------------------ rank-1 members
    def adopt_rank1(self, hyper_module):
        """hyper_module: a module with hyper_named_tensors() and a `rank1_factors` slot (hypernet_attention.HyperNet,
        hypernet.HyperNet, baseline.caption.HyperInputProj).  Its second-layer weights then never get a dense gradient: the
        backward hands over (dtheta_i, a_i), step() forms the product on the fly."""
        sink = {}
        hyper_module.rank1_factors = sink
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        for name, t in hyper_module.hyper_named_tensors().items():
            if name.endswith(".2.weight") and id(t) in mine:
                self._rank1[id(t)] = (sink, name)
        return self

    # ------------------------------------------------------------------ torch.optim.Optimizer surface
    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    ops.zero_(p.grad.view(-1))
        for sink, _ in self._rank1.values():
            sink.clear()

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dense, rank = [], []         # (group, p, grad) / (group, p, g factor, a factor)
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise CaphnError("FusedAdam implements plain Adam (weight_decay = 0, no amsgrad, no maximize), as the reference uses it")
            for p in group["params"]:
                r1 = self._rank1.get(id(p))
                fac = r1[0].pop(r1[1], None) if r1 is not None else None
                if fac:
                    if isinstance(fac, tuple):
                        fac = [fac]
                    if len(fac) > 8:
                        raise CaphnError("more than 8 accumulated backward passes for one rank-1 member")
                    if len(fac) == 1:
                        rank.append((group, p, fac[0][0].reshape(1, -1), fac[0][1].reshape(1, -1)))
                    else:       # gradient accumulation: sum_r g_r (x) a_r is a rank-R member
                        rank.append((group, p, torch.stack([f[0].reshape(-1) for f in fac]), torch.stack([f[1].reshape(-1) for f in fac])))
                elif p.grad is not None:
                    if p.grad.is_sparse:
                        raise CaphnError("FusedAdam does not take sparse gradients")
                    if not p.data.is_contiguous():
                        raise CaphnError("FusedAdam needs contiguous parameters")
                    dense.append((group, p, p.grad if p.grad.is_contiguous() else p.grad.contiguous()))
        if not dense and not rank:
            return loss
        coef = self._one
        if self.max_norm is not None:
            if len(rank) > 8:
                raise CaphnError("more than 8 rank-1 members in one clipped step")
            # (the Gram-matrix kernel takes one R for all members: members with fewer terms are padded by the caller -- here
            #  every member of a step has seen the same number of backward passes)
            if len({gf.shape[0] for _, _, gf, _ in rank}) > 1:
                raise CaphnError("rank-1 members accumulated different numbers of backward passes")
            coef, self._norm_ws = ops.grad_norm_multi([g.view(-1) for _, _, g in dense], [(gf, af) for _, _, gf, af in rank],
                                                      float(self.max_norm), 1.0, self._coef, self._norm_ws)
            self.grad_norm = coef
        # dense members: one multi-tensor launch per (hyper-parameters, step count) bucket -- normally a single one
        buckets = {}
        for group, p, g in dense:
            st = self._state_of(p)
            st["step"] += 1
            key = (group["lr"], tuple(group["betas"]), group["eps"], int(st["step"]))
            buckets.setdefault(key, []).append((p.data.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), g.view(-1)))
        for (lr, betas, eps, t), items in buckets.items():
            ops.adam_multi([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], [i[3] for i in items],
                           coef, lr, t, betas, eps)
        for group, p, gf, af in rank:
            if not p.data.is_contiguous():
                raise CaphnError("FusedAdam needs contiguous parameters")
            st = self._state_of(p)
            st["step"] += 1
            ops.adam_rank(p.data, st["exp_avg"], st["exp_avg_sq"], gf, af, coef, group["lr"], int(st["step"]),
                          group["betas"], group["eps"])
        return loss

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> dict:
        """torch.optim.Adam's layout (a torch.optim.Adam over the same parameter list loads it, and the reverse) plus `max_norm`."""
        sd = super().state_dict()
        sd["max_norm"] = self.max_norm
        return sd

    def load_state_dict(self, sd: dict) -> None:
        sd = dict(sd)
        had = "max_norm" in sd
        mn = sd.pop("max_norm", None)
        sd.pop("param_names", None)          # caphn.engine trainers add it; torch's loader does not know the key
        super().load_state_dict(sd)
        if had:
            self.max_norm = mn
        for st in self.state.values():       # torch keeps `step` on the host for non-capturable Adam; so do we
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32)

    # ------------------------------------------------------------------ conveniences kept from the first version
    @property
    def step_count(self) -> int:
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        return max(steps) if steps else 0
