"""torch.optim.Adam's update on libcaphn (caphn_adam_dense_f32), for models that run through the module API
(CATR: cc_train_catr.py:56-61 builds Adam over every parameter).  Same constructor surface for the arguments the
reference uses (params, lr, betas, eps); state lives in plain tensors next to each parameter."""
from typing import Iterable, Optional

import torch

from . import ops
from ._lib import CaphnError


class FusedAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: Optional[float] = None):
        """max_norm: torch.nn.utils.clip_grad_norm_(params, max_norm) over all gradients before the update."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedAdam got no trainable parameters")
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32:
                raise CaphnError("FusedAdam needs fp32 parameters on a CUDA(HIP) device (there is no CPU fallback)")
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.step_count = 0
        dev = self.params[0].device
        self.m = [torch.zeros_like(p.data) for p in self.params]
        self.v = [torch.zeros_like(p.data) for p in self.params]
        self._one = torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev)
        self._coef = torch.zeros(2, dtype=torch.float32, device=dev)
        self._rank1 = []         # (parameter index, factor dict, key)

    def adopt_rank1(self, hyper_module):
        """hyper_module: a module with hyper_named_tensors() and a `rank1_factors` slot (baseline.caption.HyperInputProj).  Its
        second-layer weights W2_i [w_i, k_i] then never get a dense gradient: the backward hands over (dtheta_i, a_i) and
        step() applies Adam with the rank-1 product formed on the fly (caphn_adam_rank_f32: 24 instead of 32+ bytes per
        parameter and no 4 B/parameter gradient write)."""
        if self.max_norm is not None:
            raise NotImplementedError("rank-1 gradients with global-norm clipping: use caphn.engine's trainers")
        sink = {}
        hyper_module.rank1_factors = sink
        ids = {id(p): i for i, p in enumerate(self.params)}
        for name, t in hyper_module.hyper_named_tensors().items():
            if name.endswith(".2.weight") and id(t) in ids:
                self._rank1.append((ids[id(t)], sink, name))

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                ops.zero_(p.grad.view(-1))

    def step(self):
        self.step_count += 1
        live = [(p, m, v) for p, m, v in zip(self.params, self.m, self.v) if p.grad is not None]
        coef = self._one
        if self.max_norm is not None:
            parts = torch.cat([ops.sumsq_partials(p.grad.contiguous().view(-1)) for p, _, _ in live])
            coef = ops.clip_coef(parts, None, self.max_norm, 1.0, out=self._coef)
        for i, sink, key in self._rank1:
            fac = sink.pop(key, None)
            if fac is None:
                continue
            p = self.params[i]
            ops.adam_rank(p.data, self.m[i], self.v[i], fac[0].reshape(1, -1), fac[1].reshape(1, -1), coef, self.lr, self.step_count,
                          self.betas, self.eps)
        for p, m, v in live:
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if not p.data.is_contiguous():
                raise CaphnError("FusedAdam needs contiguous parameters")
            ops.adam_dense(p.data.view(-1), m.view(-1), v.view(-1), g.view(-1), coef, self.lr, self.step_count, self.betas, self.eps)

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> dict:
        """torch.optim.Adam's state_dict layout (state[i] = {step, exp_avg, exp_avg_sq} for the i-th parameter given to the
        constructor; parameters that never received a gradient have no entry, as in torch)."""
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(zip(self.m, self.v)):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().clone(),
                            "exp_avg_sq": v.detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group], "max_norm": self.max_norm}

    def load_state_dict(self, sd: dict) -> None:
        st = sd["state"]
        steps = set()
        for i, (m, v) in enumerate(zip(self.m, self.v)):
            if i not in st:
                continue
            m.copy_(st[i]["exp_avg"].to(device=m.device, dtype=torch.float32).view_as(m))
            v.copy_(st[i]["exp_avg_sq"].to(device=v.device, dtype=torch.float32).view_as(v))
            steps.add(int(st[i]["step"]))
        if len(steps) > 1:
            raise CaphnError(f"per-parameter step counts differ ({sorted(steps)}): FusedAdam keeps one step counter")
        self.step_count = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        if "max_norm" in sd:
            self.max_norm = sd["max_norm"]
