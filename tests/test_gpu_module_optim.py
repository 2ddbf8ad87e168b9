"""The unchanged-driver path: training_step() -> loss.backward() -> (clip) -> optimizer.step() with the optimiser that
configure_optimizers() returns -- caphn.optim.FusedAdam, a torch.optim.Optimizer on libcaphn with the hypernet's second-layer
weights updated from their rank-1 factors (cc_train_hypernet.py:110-122, :395-406; hypernet_attention.py:123-133).  Against the
oracle's train_step (forward_backward -> clip_grad_norm_ -> Adam over the reference's parameter list), three steps."""
import pytest
import torch

from helpers import maxdiff
from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _V:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4, "humorous": 5, "romantic": 6}

    def __call__(self, w):
        return self.w2i.get(w, 3)

    def __len__(self):
        return len(self.w2i)


def _cc_net(dims, domains, mode="one hot"):
    from cc_train_hypernet import HyperNetCC
    from models.decoderlstm import AttentionGru
    torch.manual_seed(0)
    net = HyperNetCC(dims.F, dims.E, dims.H, dims.V, _V(), domains, lr=1e-3, hyper_emb=dims.he, embedding=mode)
    net.hypernet.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    return net.to(DEV)


def _oracle_params(net):
    p = {k[len("hypernet."):]: v.detach().cpu().clone() for k, v in net.state_dict().items() if k.startswith("hypernet.")}
    return {k: v for k, v in p.items() if not k.startswith("captioner.gru.") and not k.startswith("image_encoder.")}


@pytest.mark.parametrize("max_norm", [0.5, None])
def test_training_step_backward_fused_optimizer_matches_the_oracle(max_norm):
    from caphn.optim import FusedAdam
    domains = ["news\n", "sport\n", "travel\n"]
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=3)
    net = _cc_net(dims, domains)
    p = _oracle_params(net)
    (opt,), (sched,) = net.configure_optimizers()
    assert isinstance(opt, FusedAdam) and isinstance(opt, torch.optim.Optimizer)
    assert isinstance(sched["scheduler"], torch.optim.lr_scheduler.ReduceLROnPlateau)
    net.configure_gradient_clipping(opt, gradient_clip_val=max_norm, gradient_clip_algorithm="norm")     # what Lightning calls
    assert opt.max_norm == max_norm
    seq = ["sport", "news", "sport"]
    state = {}
    for i, d in enumerate(seq):
        b = O.synth_batch(dims, B=3, T=6, P=5, seed=20 + i)
        batch = (b["features"].to(DEV), b["captions"].to(DEV).float(), None, (d,))       # cc_dataloader.py:214-226: caps are float
        opt.zero_grad()
        loss = net.training_step(batch, i)
        loss.backward()
        w2 = net.hypernet.hn_heads[0][2].weight
        assert w2.grad is None                                    # never materialised: the factors went to the optimiser
        assert net.hypernet.hn_heads[0][0].weight.grad is not None
        opt.step()
        x = torch.zeros(dims.he); x[net.dict_domain[d]] = 1.0
        ref, *_ = O.train_step(dims, p, state, i + 1, x, b["features"], b["captions"], lr=1e-3,
                               max_norm=max_norm if max_norm is not None else 1e30)
        assert abs(float(loss) - float(ref)) < 2e-5, (i, float(loss), float(ref))
    sd = dict(net.hypernet.named_parameters())
    for n in O.trainable_names(p):
        tol = 3.5e-3 if n.endswith("v_a.bias") else 3e-5          # Adam on a mathematically zero gradient: +-lr per step
        assert maxdiff(sd[n].detach().cpu(), p[n]) < tol, n
    assert opt.step_count == 3
    if max_norm is not None:
        assert float(opt.grad_norm[1]) > max_norm                 # clipping was active (coefficient < 1)


def test_reduce_lr_on_plateau_changes_the_lr_the_kernels_use():
    """cc_train_hypernet.py:121: ReduceLROnPlateau(optimizer, cooldown=2, factor=0.5) wraps the optimiser; after the scheduler
    halves the lr, the next update is exactly half as large (Adam's first-step update is lr * sign(g))."""
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=3)

    def run(plateau):
        net = _cc_net(dims, ["a\n", "b\n", "c\n"])
        (opt,), (sched,) = net.configure_optimizers()
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=0, factor=0.5)
        if plateau:
            sch.step(1.0); sch.step(1.0)                          # no improvement -> lr halves
            assert opt.param_groups[0]["lr"] == pytest.approx(5e-4)
        b = O.synth_batch(dims, B=3, T=6, P=5, seed=7)
        before = {n: t.detach().clone() for n, t in net.hypernet.named_parameters()}
        opt.zero_grad()
        net.training_step((b["features"].to(DEV), b["captions"].to(DEV).float(), None, ("a",)), 0).backward()
        opt.step()
        return {n: (t.detach() - before[n]) for n, t in net.hypernet.named_parameters() if not n.startswith("captioner.gru.")}
    full, half = run(False), run(True)
    for n in ("hn_heads.0.2.weight", "hn_base.0.weight", "captioner.fc.weight", "captioner.embed.weight"):
        assert float(full[n].abs().max()) > 0
        assert maxdiff(half[n] * 2, full[n]) < 5e-7, n          # (p - lr u rounds in fp32: not exactly half)


def test_state_dict_interchanges_with_torch_adam():
    from caphn.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(9, 5), (17,), (4, 4)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(q.detach().clone()) for q in pa]
    oa, ob = FusedAdam(pa, lr=1e-2), torch.optim.Adam(pb, lr=1e-2)
    grads = [[torch.randn(s, generator=g).to(DEV) for s in shapes] for _ in range(4)]
    for gs in grads[:2]:
        for x, y, gr in zip(pa, pb, gs):
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    # cross-load: torch's state into the fused optimiser and the reverse
    sa, sb = oa.state_dict(), ob.state_dict()
    assert set(sa["state"][0]) == set(sb["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    oa2 = FusedAdam(pa, lr=1.0); oa2.load_state_dict(sb)
    ob2 = torch.optim.Adam(pb, lr=1.0); ob2.load_state_dict({k: v for k, v in sa.items() if k != "max_norm"})
    for gs in grads[2:]:
        for x, y, gr in zip(pa, pb, gs):
            x.grad, y.grad = gr.clone(), gr.clone()
        oa2.step(); ob2.step()
    for x, y in zip(pa, pb):
        assert maxdiff(x.detach().cpu(), y.detach().cpu()) < 2e-6


def test_gradient_accumulation_makes_a_rank_r_member():
    """Two backward passes before one step: the hypernet second-layer gradient is g1 (x) a1 + g2 (x) a2; against the dense path
    (caphn.config.TORCH_OPTIMIZER: torch.optim.Adam over materialised gradients)."""
    from caphn import config
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=3)
    out = {}
    for dense in (False, True):
        config.TORCH_OPTIMIZER = dense
        try:
            net = _cc_net(dims, ["a\n", "b\n", "c\n"])
            (opt,), _ = net.configure_optimizers()
        finally:
            config.TORCH_OPTIMIZER = False
        assert isinstance(opt, torch.optim.Adam) == dense
        opt.zero_grad()
        for i, d in enumerate(("a", "c")):
            b = O.synth_batch(dims, B=3, T=6, P=5, seed=30 + i)
            net.training_step((b["features"].to(DEV), b["captions"].to(DEV).float(), None, (d,)), i).backward()
        if dense:
            torch.nn.utils.clip_grad_norm_([q for q in opt.param_groups[0]["params"]], 0.7)
        else:
            opt.max_norm = 0.7
        opt.step()
        out[dense] = {n: t.detach().clone() for n, t in net.hypernet.named_parameters() if not n.startswith("captioner.gru.")}
    for n in out[True]:
        tol = 2.5e-3 if n.endswith("v_a.bias") else 3e-6
        assert maxdiff(out[False][n], out[True][n]) < tol, n
